"""Cuts on the GPU (DESIGN.md 4.2 "cuts"): the walk that starts at a block's cut must give the frame the walk from the
TLAS root gives (RT_NO_CUT=1: the reference's own walk, cpp/rtrayintersection.cpp:197-228), bit for bit -- hit ids,
hit t, intensity, P6 payload -- and the cut kernel must write what its host twin writes, word for word.
(tests/conftest.py makes every frame of the suite get cuts; tests/test_cuts_host.py checks the host twin against a
brute-force walk.)
"""
import numpy as np
import pytest

import raytracer_amd as ra
from raytracer_amd import meshgen

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float64).view(np.uint64)


@pytest.fixture(scope="module")
def ctx():
    c = ra.Context(0)
    yield c
    c.close()


def both_ways(ctx, monkeypatch, cam, H, W, spp, **kw):
    """The same frame with and without cuts (same stream position / same explicit offsets)."""
    out = {}
    for mode in ("cut", "root"):
        if mode == "root":
            monkeypatch.setenv("RT_NO_CUT", "1")
        else:
            monkeypatch.delenv("RT_NO_CUT", raising=False)
        ctx.rng_seed(5489)
        ctx.render(*cam, H, W, spp, hit_ids=True, hit_t=True, **kw)
        out[mode] = ctx.download()
        n_blocks = ctx.last_cuts()[0].shape[0]
        assert (n_blocks > 0) == (mode == "cut"), (mode, n_blocks)
    monkeypatch.delenv("RT_NO_CUT", raising=False)
    a, b = out["cut"], out["root"]
    assert np.array_equal(a["hit_prim"], b["hit_prim"])
    assert np.array_equal(bits(a["hit_t"]), bits(b["hit_t"]))
    assert np.array_equal(bits(a["intensity"]), bits(b["intensity"]))
    assert np.array_equal(a["rgb"], b["rgb"])
    return a


def scenes():
    c0, k0 = meshgen.plate(24, T=1)
    c1, k1 = meshgen.uv_sphere(10, 14, radius=0.8, center=(0.4, 0.3, -2.5))
    yield "plate+sphere", [c0, c1], [k0, k1]
    c, k = meshgen.random_soup(3000, seed=7, extent=4.0, z_range=(-6.0, -2.0))
    yield "soup", [c], [k]
    rng = np.random.default_rng(5)
    cs, ks = [], []
    for m in range(8):  # many small meshes: the TLAS slot quirk (rtbvh.cpp:552-560) decides what a leaf guards
        c, k = meshgen.plate(10, side=2.5, x0=float(rng.uniform(-5, 2.5)), y0=float(rng.uniform(-4, 1.5)),
                             z0=float(rng.uniform(-9, -4)))
        cs.append(c)
        ks.append(k)
    yield "8 plates", cs, ks
    c, k = meshgen.random_soup(1500, seed=9, extent=3.0, z_range=(-3.0, 3.0))
    yield "cloud around the camera", [c], [k]


CAMS = [dict(), dict(center=(0.3, 0.1, 0.2), target=(0.0, 0.0, -2.0), fov=70),
        dict(center=(2.5, 1.5, 1.0), target=(0.4, 0.3, -2.5), fov=50), dict(center=(0.0, 0.0, 0.0), target=(0.3, 1.0, -0.2), fov=100)]


@pytest.mark.parametrize("name,coords,colors", list(scenes()), ids=[s[0] for s in scenes()])
def test_cut_walk_equals_root_walk(ctx, monkeypatch, name, coords, colors):
    host = ra.HostScene(coords, colors, 0)
    ctx.upload(host)
    shapes = [(40, 56, 64), (37, 50, 50), (24, 33, 5), (21, 19, 1), (16, 24, 128), (30, 40, 100)]
    hits = 0
    for ci, ck in enumerate(CAMS):
        H, W, spp = shapes[ci % len(shapes)] if name != "soup" else shapes[(ci + 2) % len(shapes)]
        cam = meshgen.camera_arrays(W, H, **ck)
        a = both_ways(ctx, monkeypatch, cam, H, W, spp)
        hits += int((a["hit_prim"] >= 0).sum())
    assert hits > 0
    host.close()


def test_cut_walk_equals_root_walk_row_bands_and_all_sample_counts(ctx, monkeypatch):
    c0, k0 = meshgen.plate(40)
    host = ra.HostScene([c0], [k0], 0)
    ctx.upload(host)
    H, W = 48, 72
    cam = meshgen.camera_arrays(W, H, center=(0.2, -0.1, 0.0), target=(0.0, 0.1, -1.0), fov=85)
    for spp in (1, 2, 3, 4, 7, 16, 31, 32, 33, 63, 64, 65, 256):
        both_ways(ctx, monkeypatch, cam, H, W, spp)
    for rows in ((0, 9), (9, 30), (30, 48)):  # the multi-GPU row-band schedule: blocks are relative to the band
        both_ways(ctx, monkeypatch, cam, H, W, 64, rows=rows)
        both_ways(ctx, monkeypatch, cam, H, W, 50, rows=rows)
    host.close()


def test_offsets_outside_the_pixel_fall_back_to_the_root_walk(ctx, monkeypatch):
    """RT_SAMPLER_EXPLICIT takes any offsets; a cut only covers offsets in [-0.5, 0.5]: batches with others must
    walk from the root (same frame either way, and the same as all-in-range offsets where those are used)."""
    c0, k0 = meshgen.plate(30)
    host = ra.HostScene([c0], [k0], 0)
    ctx.upload(host)
    H, W, spp = 24, 32, 64
    cam = meshgen.camera_arrays(W, H)
    rng = np.random.default_rng(3)
    jit = rng.uniform(-0.5, 0.5, (H * W * spp, 2))
    wild = jit.copy()
    sel = rng.random(H * W * spp) < 0.02
    wild[sel] = rng.uniform(-3.0, 3.0, (int(sel.sum()), 2))  # samples that land in other pixels' blocks
    wild[5] = (0.5, -0.5)  # exactly on the edge: still covered
    a = both_ways(ctx, monkeypatch, cam, H, W, spp, jitter=jit)
    b = both_ways(ctx, monkeypatch, cam, H, W, spp, jitter=wild)
    same = ~sel
    assert np.array_equal(a["hit_prim"][same], b["hit_prim"][same])
    host.close()


def test_gpu_cuts_equal_the_host_twin_word_for_word(ctx, monkeypatch):
    c0, k0 = meshgen.plate(24)
    c1, k1 = meshgen.uv_sphere(10, 14, radius=0.8, center=(0.4, 0.3, -2.5))
    host = ra.HostScene([c0, c1], [k0, k1], 0)
    ctx.upload(host)
    for (H, W, spp), ck in zip([(40, 56, 64), (37, 50, 5), (21, 19, 1)], CAMS):
        cam = meshgen.camera_arrays(W, H, **ck)
        ctx.render(*cam, H, W, spp)
        _, _, (lo, top) = ctx.last_cuts()
        for level in sorted({lo, top}):
            got, tx, _ = ctx.last_cuts(level)
            want, tx2, (lo2, top2) = ra.tile_cuts_host(host, *cam, H, W, spp, level=level)
            assert (tx, lo, top) == (tx2, lo2, top2) and got.shape == want.shape and got.shape[0] > 0
            assert np.array_equal(got[:, 0], want[:, 0])
            n = got[:, 0] & 0xff
            live = np.arange(31)[None, :] < np.where(got[:, 0] >> 31, n, 0)[:, None]
            assert np.array_equal(got[:, 1:][live], want[:, 1:][live])
    host.close()


def test_small_frames_get_no_cuts_by_default(ctx, monkeypatch):
    """The cut kernels cost ~0.3 ms whatever the image: by default only frames of >= 12 M rays get them."""
    monkeypatch.delenv("RT_CUT_MIN_RAYS")
    c0, k0 = meshgen.plate(16)
    host = ra.HostScene([c0], [k0], 0)
    ctx.upload(host)
    cam = meshgen.camera_arrays(64, 48)
    ctx.render(*cam, 48, 64, 64)
    assert ctx.last_cuts()[0].shape[0] == 0
    monkeypatch.setenv("RT_CUT_MIN_RAYS", "0")
    ctx.render(*cam, 48, 64, 64)
    assert ctx.last_cuts()[0].shape[0] == 24 * 32
    host.close()


@pytest.mark.parametrize("name,coords,colors", list(scenes()), ids=[s[0] for s in scenes()])
def test_single_level_cut_walk_equals_root_walk(ctx, monkeypatch, name, coords, colors):
    """The one-level mode of mid-sized frames (RT_CUT1_MIN_RAYS=0: for every frame here), every scene, camera, tile
    size and instantiation the full hierarchy is tested with."""
    monkeypatch.delenv("RT_CUT_MIN_RAYS")
    monkeypatch.setenv("RT_CUT1_MIN_RAYS", "0")
    monkeypatch.setenv("RT_CUT1_MAX_TRIS_PER_BLOCK", "1e9")  # (these images are a few hundred blocks)
    host = ra.HostScene(coords, colors, 0)
    ctx.upload(host)
    shapes = [(40, 56, 64), (37, 50, 50), (24, 33, 5), (21, 19, 1), (16, 24, 128), (30, 40, 100), (33, 47, 16), (90, 120, 3)]
    for ci, ck in enumerate(CAMS + CAMS):
        H, W, spp = shapes[(ci + (3 if name == "soup" else 0)) % len(shapes)]
        cam = meshgen.camera_arrays(W, H, **ck)
        both_ways(ctx, monkeypatch, cam, H, W, spp)
        _, _, (lo, top) = ctx.last_cuts()
        assert lo == top
        both_ways(ctx, monkeypatch, cam, H, W, spp, rows=(8, H - 3))
    host.close()


def test_mid_sized_frames_get_one_short_cut_level_at_the_render_tile(ctx, monkeypatch):
    """Frames of 1 M .. 12 M rays (the reference's own 400 x 225 x 50): ONE cut level, the render tile's blocks,
    straight from the root by the breadth-first kernel (cut_single_kernel), leaves trimmed -- and the same frame as
    the root walk."""
    monkeypatch.delenv("RT_CUT_MIN_RAYS")
    c0, k0 = meshgen.plate(24)
    c1, k1 = meshgen.uv_sphere(10, 14, radius=0.8, center=(0.4, 0.3, -2.5))
    host = ra.HostScene([c0, c1], [k0, k1], 0)
    ctx.upload(host)
    for (H, W, spp), ck in zip([(100, 160, 64), (150, 280, 50), (225, 400, 50)], CAMS):  # 1.02 M, 2.1 M, 4.5 M rays
        cam = meshgen.camera_arrays(W, H, **ck)
        a = both_ways(ctx, monkeypatch, cam, H, W, spp)
        assert (a["hit_prim"] >= 0).any()
        ctx.rng_seed(5489)
        ctx.render(*cam, H, W, spp)
        words, tx, (lo, top) = ctx.last_cuts()
        assert lo == top and ctx.stats()["finest_cut_level"] == lo
        assert words.shape[0] == tx * ((H + (1 << lo) - 1) >> lo) and tx == (W + (1 << lo) - 1) >> lo
        valid = (words[:, 0] >> 31) == 1
        assert valid.mean() > 0.9 and ((words[valid, 0] & 0xff) > 0).any()
        # the breadth-first kernel (one lane per entry) against its serial host twin, word for word
        want, tx2, (lo2, top2) = ra.tile_cuts_host(host, *cam, H, W, spp, level=lo)
        assert (tx, lo, top) == (tx2, lo2, top2) and np.array_equal(words[:, 0], want[:, 0])
        n = words[:, 0] & 0xff
        live = np.arange(31)[None, :] < np.where(valid, n, 0)[:, None]
        assert np.array_equal(words[:, 1:][live], want[:, 1:][live])
    # below 1 M rays: none; more than 8 triangles per block: none; from 12 M rays: the whole hierarchy (several levels)
    cam = meshgen.camera_arrays(200, 150)
    ctx.render(*cam, 150, 200, 32)
    assert ctx.last_cuts()[0].shape[0] == 0
    monkeypatch.setenv("RT_CUT1_MAX_TRIS_PER_BLOCK", "0.5")
    ctx.render(*cam, 150, 200, 64)
    assert ctx.last_cuts()[0].shape[0] == 0
    monkeypatch.delenv("RT_CUT1_MAX_TRIS_PER_BLOCK")
    ctx.render(*cam, 150, 200, 64)
    assert ctx.last_cuts()[0].shape[0] > 0
    monkeypatch.setenv("RT_CUT_MIN_RAYS", "0")
    ctx.render(*cam, 150, 200, 64)
    _, _, (lo, top) = ctx.last_cuts()
    assert top > lo
    host.close()


def test_triangle_edges_on_block_corners_and_samples_on_edges(ctx, monkeypatch):
    """Adversarial for the trimmed leaves (cut_trim): a flat plate whose quad edges project EXACTLY onto pixel
    corners (every second one, so also onto the corners of the 2 x 2 blocks), a camera whose pixel grid is made of
    binary fractions, and explicit sample offsets from {-0.5, 0, +0.5} (rays through edges and vertices) and within
    2^-30 of them.  The block-level test must keep every triangle such a ray can hit: cut walk == root walk."""
    H, W, spp = 32, 48, 64
    d = 2.0 ** -5
    cam_c = np.zeros(3)
    p00 = np.array([-(W - 1) / 2 * d, (H - 1) / 2 * d, -1.0])
    spacing = np.array([[d, 0.0, 0.0], [0.0, -d, 0.0]])
    # plate at z = -2: a vertex at 2 * (pixel corner) projects onto that corner; quads of 2 x 2 pixels
    z = -2.0
    xs = 2.0 * (p00[0] + (np.arange(0, W + 1, 2) - 0.5) * d)
    ys = 2.0 * (p00[1] - (np.arange(0, H + 1, 2) - 0.5) * d)
    tris = []
    for i in range(len(xs) - 1):
        for j in range(len(ys) - 1):
            a, b = (xs[i], ys[j + 1], z), (xs[i + 1], ys[j + 1], z)
            c, e = (xs[i + 1], ys[j], z), (xs[i], ys[j], z)
            tris += [(a, b, c), (a, c, e)]  # counter-clockwise seen from the camera (+z side)
    coords = np.array(tris, np.float64)[None]
    colors = np.linspace(0.1, 0.9, coords.shape[1])[None, :, None].repeat(3, axis=2)
    host = ra.HostScene([coords], [colors], 0)
    ctx.upload(host)
    rng = np.random.default_rng(17)
    n = H * W * spp
    base = rng.choice([-0.5, 0.0, 0.5], size=(n, 2))
    nudge = rng.choice([0.0, 2.0 ** -30, -(2.0 ** -30), 2.0 ** -52], size=(n, 2))
    jit = np.clip(base + nudge, -0.5, 0.5)
    a = both_ways(ctx, monkeypatch, (cam_c, p00, spacing), H, W, spp, jitter=jit)
    assert (a["hit_prim"] >= 0).mean() > 0.9
    # and with the sampler's own offsets
    both_ways(ctx, monkeypatch, (cam_c, p00, spacing), H, W, spp)
    both_ways(ctx, monkeypatch, (cam_c, p00, spacing), H, W, 50)
    host.close()
