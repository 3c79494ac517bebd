"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the same seeded
inputs.  Bars (BASELINE.json north_star): per-sample hit triangle bit-exact, pixel intensity within 1e-6
abs -- we additionally assert bit-equality of intensities, t, barycentrics and the P6 payload, which the
fp64 no-FMA kernel is designed to deliver.  Run with -m gpu on an MI355X."""
import os

import numpy as np
import pytest

import orc
import raytracer_amd as ra
from raytracer_amd import meshgen

pytestmark = pytest.mark.gpu

INTENSITY_TOL = 1e-6  # north_star tolerance; tests also check exact equality


def bits(a):
    return np.ascontiguousarray(a, np.float64).view(np.uint64)


@pytest.fixture(scope="module")
def ctx():
    c = ra.Context(0)
    yield c
    c.close()


def check_image(ctx, coords, colors, cam, H, W, spp, timestep=0, jitter=None, seed_pos=0):
    host = ra.HostScene(coords, colors, timestep)
    ctx.upload(host)
    oscene = orc.Scene(coords, colors, timestep)
    cc, p00, sp = cam
    if jitter is None:
        ctx.rng_seed(5489)
        ctx.rng_seek(seed_pos)
        rng = orc.Rng(5489)
        rng.discard(seed_pos)
        want = oscene.render(cc, p00, sp, H, W, spp, rng=rng)
    else:
        want = oscene.render(cc, p00, sp, H, W, spp, jitter=jitter)
    ctx.render(cc, p00, sp, H, W, spp, jitter=jitter, hit_ids=True, hit_t=True)
    got = ctx.download()
    if jitter is None:
        assert ctx.rng_tell() == seed_pos + H * W * spp * 4
    mesh, elem = host.map_prims(got["hit_prim"])
    assert np.array_equal(mesh, want["hit_mesh"]), "hit mesh ids differ"
    assert np.array_equal(elem, want["hit_elem"]), "hit element ids differ"
    assert np.array_equal(bits(got["hit_t"]), bits(want["hit_t"]))
    assert np.max(np.abs(got["intensity"] - want["intensity"])) <= INTENSITY_TOL
    assert np.array_equal(bits(got["intensity"]), bits(want["intensity"]))
    assert np.array_equal(got["rgb"], want["rgb"])
    return got, want, host


def test_jitter_stream_matches_reference_rng(ctx):
    for n, streams in [(10, 1), (624, 1), (5000, 3), (100000, 7), (312 * 8, 4)]:
        ctx.rng_seed(5489)
        got = ctx.rng_jitter(n, streams)
        want = orc.Rng(5489).doubles(n) - 0.5
        assert np.array_equal(bits(got), bits(want)), (n, streams)
        assert ctx.rng_tell() == 2 * n
    # continuation + seek
    ctx.rng_seed(5489)
    a = ctx.rng_jitter(1000, 2)
    b = ctx.rng_jitter(1000, 5)
    want = orc.Rng(5489).doubles(2000) - 0.5
    assert np.array_equal(bits(np.concatenate([a, b])), bits(want))
    ctx.rng_seek(2 * 777)
    c = ctx.rng_jitter(100, 2)
    assert np.array_equal(bits(c), bits(want[777:877]))


def test_trace_rays_random_soup(ctx):
    c, k = meshgen.random_soup(3000, seed=7)
    host = ra.HostScene([c], [k])
    ctx.upload(host)
    o = orc.Scene([c], [k])
    rng = np.random.default_rng(3)
    n = 20000
    org = np.zeros((n, 3)) + rng.uniform(-0.5, 0.5, (n, 3))
    d = np.stack([rng.uniform(-1, 1, n), rng.uniform(-1, 1, n), -np.ones(n)], -1)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    got = ctx.trace(org, d)
    want = o.trace(org, d)
    mesh, elem = host.map_prims(got["prim"])
    assert (want["elem"] >= 0).sum() > 2000
    assert np.array_equal(elem, want["elem"]) and np.array_equal(mesh, want["mesh"])
    assert np.array_equal(bits(got["t"]), bits(want["t"]))
    assert np.array_equal(bits(got["bary"]), bits(want["bary"]))
    assert np.array_equal(bits(got["color"]), bits(want["color"]))


def test_trace_axis_aligned_rays_nan_slab_order(ctx):
    """Direction components exactly +-0 and origins on box planes: 0*inf = NaN inside the slab test; the
    kernel must follow std::min/std::max operand order (rtrayintersection.cpp:136-138)."""
    tris = []
    for ix in range(-3, 3):
        for iy in range(-3, 3):
            z = -2.0 - ((ix + iy) % 3)
            tris.append([[ix, iy, z], [ix + 1, iy, z], [ix, iy + 1, z]])
            tris.append([[ix + 1, iy, z], [ix + 1, iy + 1, z], [ix, iy + 1, z]])
    c = np.array(tris, float)[None]
    k = np.linspace(0.1, 0.9, c.shape[1])[None, :, None].repeat(3, 2)
    host = ra.HostScene([c], [k])
    ctx.upload(host)
    o = orc.Scene([c], [k])
    org, d = [], []
    for x in np.arange(-3, 3.01, 0.5):
        for y in np.arange(-3, 3.01, 0.5):
            for dz in (-1.0,):
                for dx, dy in ((0.0, 0.0), (-0.0, 0.0), (0.0, -0.0), (1.0, 0.0), (0.0, 1.0)):
                    org.append([x, y, 0.0])
                    v = np.array([dx, dy, dz if (dx == 0 and dy == 0) else 0.0])
                    d.append(v)
    org, d = np.array(org), np.array(d)
    got = ctx.trace(org, d)
    want = o.trace(org, d)
    mesh, elem = host.map_prims(got["prim"])
    assert np.array_equal(elem, want["elem"])
    assert np.array_equal(bits(got["t"]), bits(want["t"]))
    assert np.array_equal(bits(got["color"]), bits(want["color"]))
    assert (want["elem"] >= 0).sum() > 50


@pytest.mark.parametrize("spp", [1, 3, 4, 50, 64, 128, 192, 256])
def test_render_plate_matches_oracle(ctx, spp):
    c, k = meshgen.plate(48)
    cam = meshgen.camera_arrays(40, 24)
    got, want, _ = check_image(ctx, [c], [k], cam, 24, 40, spp)
    assert (want["hit_elem"] >= 0).mean() > 0.5


def test_render_odd_sizes_and_stream_position(ctx):
    c, k = meshgen.plate(20)
    cam = meshgen.camera_arrays(37, 19, center=(0.3, -0.2, 0.5), target=(0.1, 0.0, -1.0), fov=70)
    check_image(ctx, [c], [k], cam, 19, 37, 5, seed_pos=4 * 12345)
    cam = meshgen.camera_arrays(1, 1)
    check_image(ctx, [c], [k], cam, 1, 1, 1)
    cam = meshgen.camera_arrays(3, 70)
    check_image(ctx, [c], [k], cam, 70, 3, 2)


def _cloud(n, seed, extent=5.0, size=0.5):
    """Randomly oriented triangles all around the origin (about half face any given viewer)."""
    rng = np.random.default_rng(seed)
    c = rng.uniform(-extent, extent, (n, 3))
    c = c[np.linalg.norm(c, axis=1) > 1.0]
    arr = c[:, None, :] + rng.uniform(-size, size, (c.shape[0], 3, 3))
    g = rng.uniform(0.05, 0.95, (c.shape[0], 3))  # genuinely RGB face colours
    return np.ascontiguousarray(arr[None]), np.ascontiguousarray(g[None])


@pytest.mark.parametrize("spp", [2, 64])
def test_every_direction_octant_and_mixed_sign_waves(ctx, spp):
    """The packet kernel specialises its slab test per direction octant when a whole wave looks the same
    way: cameras towards all 8 octants (wave-uniform signs) and along the 6 axes (signs
    mixed inside a wave) must give the oracle's image."""
    c, k = _cloud(6000, 11)
    targets = [(sx, sy, sz) for sx in (-1.0, 1.0) for sy in (-0.9, 1.1) for sz in (-1.2, 0.8)]
    targets += [(1.0, 0.01, 0.02), (-1.0, 0.02, 0.01), (0.01, 0.02, 1.0), (0.02, 0.01, -1.0), (0.03, 1.0, 0.2),
                (0.01, -1.0, 0.2)]
    hits = 0
    for i, t in enumerate(targets):
        cam = meshgen.camera_arrays(24, 16, center=(0.05, -0.03, 0.02), target=t, fov=60 if i < 8 else 100)
        got, want, _ = check_image(ctx, [c], [k], cam, 16, 24, spp, seed_pos=4 * 1000 * i)
        hits += int((want["hit_elem"] >= 0).sum())
    assert hits > 14 * 16 * 24 * spp * 0.2


@pytest.mark.parametrize("n_meshes", [2, 3, 5, 8])
def test_render_multi_mesh_including_slot_quirk(ctx, n_meshes):
    rng = np.random.default_rng(50 + n_meshes)
    coords, colors = [], []
    for m in range(n_meshes):
        c, k = meshgen.uv_sphere(10, 14, radius=0.8, center=tuple(rng.uniform(-3, 3, 2)) + (-6.0 + rng.uniform(-1, 1),))
        coords.append(c)
        colors.append(k * (0.2 + 0.1 * m))
    cam = meshgen.camera_arrays(48, 32)
    got, want, _ = check_image(ctx, coords, colors, cam, 32, 48, 4)
    assert (want["hit_mesh"] >= 0).any()


def test_unjittered_rays_through_shared_vertices_ties(ctx):
    """Zero jitter sends rays exactly through grid vertices/edges of a flat plate: equal-t candidates in
    neighbouring triangles; first-found (reference traversal order) must win."""
    c, k = meshgen.plate(8, side=8.0, z0=-4.0, amp=0.0)
    H = W = 9
    cam = (np.zeros(3), np.array([-4.0, 4.0, -4.0]), np.array([[1.0, 0, 0], [0, -1.0, 0]]))
    jit = np.zeros((H * W * 1, 2))
    got, want, _ = check_image(ctx, [c], [k], cam, H, W, 1, jitter=jit)
    assert (want["hit_elem"] >= 0).sum() >= 60


def test_duplicate_meshes_and_big_leaves(ctx):
    # nine coincident triangles (one oversized leaf -> leaf_ext path) twice (two coincident meshes)
    base = np.array([[-2.0, -2, -3], [2, -2, -3], [0, 2, -3]])
    c = np.repeat(base[None], 9, axis=0)[None]
    k = (np.arange(9)[None, :, None] / 10.0 + 0.05).repeat(3, 2)
    cam = meshgen.camera_arrays(16, 12)
    got, want, host = check_image(ctx, [c, c.copy()], [k, k * 0.5], cam, 12, 16, 2)
    assert host.info.max_leaf_tris == 9
    hit = want["hit_elem"][want["hit_elem"] >= 0]
    assert hit.size and (hit == hit[0]).all()  # the first-found triangle wins every tie


def test_deep_tree_uses_global_stack(ctx):
    # geometric spacing makes the binned split peel off few elements per level: a very deep, left-leaning tree
    tris = []
    n_t = 300
    for i in range(n_t):
        x = 3.0 ** i * 1e-6
        tris.append([[x, -1.0, -3.0], [x * 1.5, -1.0, -3.0], [x, 1.0, -3.0]])
    c = np.array(tris)[None]
    k = np.linspace(0.2, 0.8, n_t)[None, :, None].repeat(3, 2)
    host = ra.HostScene([c], [k])
    assert host.info.stack_need > 48, host.info.stack_need
    ctx.upload(host)
    o = orc.Scene([c], [k])
    rng = np.random.default_rng(0)
    n = 4000
    xi = 3.0 ** rng.integers(0, n_t, n) * 1e-6
    org = np.stack([xi * rng.uniform(0.9, 1.6, n), rng.uniform(-0.9, 0.7, n), np.zeros(n)], -1)
    d = np.tile([0.0, 0.0, -1.0], (n, 1)) + rng.uniform(-1e-4, 1e-4, (n, 3))
    got = ctx.trace(org, d)
    want = o.trace(org, d)
    _, elem = host.map_prims(got["prim"])
    assert np.array_equal(elem, want["elem"])
    assert np.array_equal(bits(got["t"]), bits(want["t"]))
    assert (want["elem"] >= 0).sum() > 1000
    cam = meshgen.camera_arrays(8, 8)
    check_image(ctx, [c], [k], cam, 8, 8, 2)


@pytest.mark.parametrize("spp", [3, 64])  # 64: the one-pixel-per-batch instantiation, with row0 != 0
def test_row_bands_equal_full_image(ctx, spp):
    c, k = meshgen.plate(32)
    H, W = 40, 24
    cam = meshgen.camera_arrays(W, H)
    host = ra.HostScene([c], [k])
    ctx.upload(host)
    ctx.rng_seed(5489)
    ctx.render(*cam, H, W, spp, hit_ids=True)
    full = ctx.download()
    parts = []
    for a, b in [(0, 7), (7, 8), (8, 33), (33, 40)]:
        ctx.rng_seek(a * W * spp * 4)
        ctx.render(*cam, H, W, spp, rows=(a, b), hit_ids=True)
        parts.append(ctx.download())
    assert np.array_equal(np.concatenate([p["rgb"] for p in parts]), full["rgb"])
    assert np.array_equal(bits(np.concatenate([p["intensity"] for p in parts])), bits(full["intensity"]))
    assert np.array_equal(np.concatenate([p["hit_prim"] for p in parts]), full["hit_prim"])


def test_counters_match_oracle_visit_counts(ctx):
    c, k = meshgen.plate(64)
    H, W, spp = 32, 32, 2
    cam = meshgen.camera_arrays(W, H)
    host = ra.HostScene([c], [k])
    ctx.upload(host)
    ctx.rng_seed(5489)
    ctx.render(*cam, H, W, spp, counters=True)
    st = ctx.stats()
    want = orc.Scene([c], [k]).render(*cam, H, W, spp, rng=orc.Rng())
    assert st["rays"] == H * W * spp
    assert st["nodes_visited"] == want["nodes"]
    assert st["tris_tested"] == want["tris"]
    assert st["rays_hit"] == want["hits"]
    assert st["render_kernel_ms"] > 0


def test_render_scene_driver_files_and_sharding(ctx, tmp_path):
    """rt_render_scene == the reference's render_scene loop (rtmain.cpp:49-64): timestep outer, camera inner,
    one never-reseeded RNG stream across frames, files rtimage_<t>_cam<c>.ppm."""
    c, k = meshgen.plate(16, T=3, sway=0.2)
    c2, k2 = meshgen.uv_sphere(8, 10, radius=1.0, center=(0.5, 0.5, -3.0), T=3)
    H, W, spp = 18, 30, 2
    cams = [meshgen.camera_arrays(W, H), meshgen.camera_arrays(W, H, center=(0.5, 0.2, 0.3), target=(0, 0, -2), fov=60)]
    sc = ra.Scene()
    sc.add_mesh(c, None, None, k, 3)
    sc.add_mesh(c2, None, None, k2, 3)
    for cc, p00, sp in cams:
        sc.add_camera(cc, p00, sp)
    out = tmp_path / "all"
    out.mkdir()
    from raytracer_amd import _lib as L
    # the driver's stream is process-global and never reseeded (rtmathutils.h:15-16): other tests of this
    # process have advanced it, so the oracle starts where the driver stands now
    base = int(L.lib.rt_driver_stream_position(0))
    ra.render_scene(H, W, sc, spp, out)
    names = sorted(os.listdir(out))
    assert names == [f"rtimage_{t}_cam{cam}.ppm" for t in range(3) for cam in range(2)]
    frame_words = H * W * spp * 4
    assert int(L.lib.rt_driver_stream_position(0)) == base + 6 * frame_words
    hdr = f"P6\n{W} {H}\n255\n".encode()
    rng = orc.Rng()
    rng.set_state(ra.mt19937_state_at(base))
    for t in range(3):
        osc = orc.Scene([c, c2], [k, k2], t)
        for ci, cam in enumerate(cams):
            want = osc.render(*cam, H, W, spp, rng=rng, want_hits=False)
            data = (out / f"rtimage_{t}_cam{ci}.ppm").read_bytes()
            assert data == hdr + want["rgb"].tobytes(), (t, ci)


_SHARD_SCRIPT = r'''
import sys, os
sys.path.insert(0, sys.argv[1])
import numpy as np
import raytracer_amd as ra
from raytracer_amd import meshgen
out, rank, world = sys.argv[2], sys.argv[3], sys.argv[4]
os.environ["RT_SHARD_FRAMES"] = "1"; os.environ["RANK"] = rank; os.environ["WORLD_SIZE"] = world
os.environ["LOCAL_RANK"] = "0"
c, k = meshgen.plate(12, T=3, sway=0.2)
sc = ra.Scene(); sc.add_mesh(c, None, None, k, 3)
for cam in (meshgen.camera_arrays(20, 12), meshgen.camera_arrays(20, 12, center=(0.4, 0.1, 0.2), fov=70)):
    sc.add_camera(*cam)
ra.render_scene(12, 20, sc, 3, out)
'''


def test_frame_sharding_gives_identical_files(tmp_path):
    """Each rank is its own process (as under torchrun); world=2 ranks together must write exactly the files a
    single process writes, byte for byte (every rank seeks the global RNG stream to its frames)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "shard.py"
    script.write_text(_SHARD_SCRIPT)
    single, split = tmp_path / "single", tmp_path / "split"
    single.mkdir()
    split.mkdir()
    subprocess.check_call([sys.executable, str(script), root, str(single), "0", "1"], timeout=300)
    for r in ("0", "1"):
        subprocess.check_call([sys.executable, str(script), root, str(split), r, "2"], timeout=300)
    names = sorted(os.listdir(single))
    assert len(names) == 6 and names == sorted(os.listdir(split))
    for n in names:
        assert (single / n).read_bytes() == (split / n).read_bytes(), n


def test_reference_sample_files_render_like_the_oracle(ctx, tmp_path):
    """BASELINE configs[0] and [1] from the reference's own data files (tests/golden/): simple_block.e over all
    9 timesteps through the render_scene driver, and the VTK sphere at 50 spp."""
    from raytracer_amd import rtsimdataloader as loader
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    # configs[1]
    m = loader.vtk_mesh_to_rtmesh(os.path.join(gold, "sphere_gmsh_vtk_test_2d_manyelems.vtk"), scale=2.0,
                                  world_position=np.array([0.0, 0.0, -2.2]))
    m.set_surface(loader.SurfType.FIELD_COLOR, np.array([0.5, 0.5, 0.5]))
    cam = meshgen.camera_arrays(80, 45)
    got, want, _ = check_image(ctx, [m.node_coords_expanded_over_time], [m.face_colors_over_time], cam, 45, 80, 50)
    assert 0.05 < (want["hit_elem"] >= 0).mean() < 0.9
    # configs[0], every timestep
    b = loader.simdata_to_rtmesh(os.path.join(gold, "simple_block.e"), scale=100.0,
                                 world_position=np.array([-0.5, -0.75, -2.0]), color_field="disp_x")
    for t in (0, 4, 8):
        got, want, _ = check_image(ctx, [b.node_coords_expanded_over_time], [b.face_colors_over_time],
                                   meshgen.camera_arrays(100, 56), 56, 100, 1, timestep=t)
        assert (want["hit_elem"] >= 0).mean() > 0.05


def test_device_jump_matches_host_jump_and_sequential(ctx):
    """The GPU jump kernels (seek, stream doubling tree) against the host GF(2) jump and sequential generation."""
    from test_mt_jump import oracle_words_from_state
    for pos in (1, 623, 624, 1000, 360000, 2**33 + 12346):
        ctx.rng_seed(5489)
        ctx.rng_seek(pos)
        dev = ctx.rng_state()
        host = ra.mt19937_state_at(pos)
        assert np.array_equal(dev[1:], host[1:]) and (dev[0] >> 31) == (host[0] >> 31), pos
        assert np.array_equal(oracle_words_from_state(dev, 700), oracle_words_from_state(host, 700))
    # far position, many streams, then a backward seek
    ctx.rng_seed(5489)
    ctx.rng_seek(2**40 + 6)
    got = ctx.rng_jitter(50000, 37)
    g = orc.Rng()
    g.set_state(ra.mt19937_state_at(2**40 + 6))
    assert np.array_equal(bits(got), bits(g.doubles(50000) - 0.5))
    assert ctx.rng_tell() == 2**40 + 6 + 100000
    ctx.rng_seek(10)
    got = ctx.rng_jitter(100, 1)
    g = orc.Rng()
    g.discard(10)
    assert np.array_equal(bits(got), bits(g.doubles(100) - 0.5))
    # other seed
    ctx.rng_seed(42)
    got = ctx.rng_jitter(1000, 3)
    assert np.array_equal(bits(got), bits(orc.Rng(42).doubles(1000) - 0.5))
    ctx.rng_seed(5489)


def test_one_frame_in_row_bands_over_three_ranks_equals_the_full_frame(ctx):
    """SURVEY 8(e), single big frame: every rank renders its band from rt_shard_rows at its own stream
    position; the assembled frame is the single-GPU frame bit for bit (ranks simulated one after another
    on this GPU) and every rank leaves the stream at the end of the frame."""
    c, k = meshgen.plate(64)
    H, W, spp, base = 70, 52, 5, 4 * 4321
    cam = meshgen.camera_arrays(W, H)
    ctx.upload(ra.HostScene([c], [k], 0))
    ctx.rng_seed(5489)
    ctx.rng_seek(base)
    ctx.render(*cam, H, W, spp)
    full = ctx.download()
    inten = np.full((H, W), np.nan)
    rgb = np.zeros((H, W, 3), np.uint8)
    for rank in (2, 0, 1):
        ctx.rng_seed(5489)
        band, (r0, r1) = ra.render_frame_banded(ctx, *cam, H, W, spp, rank, 3, frame_base=base)
        assert ctx.rng_tell() == base + H * W * spp * 4
        inten[r0:r1] = band["intensity"]
        rgb[r0:r1] = band["rgb"]
    assert np.array_equal(bits(inten), bits(full["intensity"]))
    assert np.array_equal(rgb, full["rgb"])


# ----------------------------------------------------------------------------------------------
# The HIP path against the committed golden vectors themselves (tests/golden/eigen_probe.npz: values
# produced by the reference's own rtmathutils.h and vendored Eigen, see oracle/make_golden.py).
# ----------------------------------------------------------------------------------------------
def test_gpu_jitter_equals_the_references_random_double_golden(ctx, golden_dir):
    gold = np.load(os.path.join(golden_dir, "eigen_probe.npz"))
    want = gold["rng_doubles"] - 0.5  # cpp/rtrender.cpp:53: random_double() - 0.5
    ctx.rng_seed(5489)
    got = ctx.rng_jitter(want.size, 3)
    assert np.array_equal(bits(got), bits(want))


def test_gpu_leaf_intersection_equals_eigen_golden(ctx, golden_dir):
    """Each probe is one leaf (1..4 triangles) and one ray evaluated by the reference's Eigen expressions
    (cpp/rtrayintersection.cpp:36-122 shapes): t, u, v, w per triangle and minCoeff's first-minimum row.
    The same leaf as a one-mesh scene through rt_trace_rays must give that row and those bits."""
    gold = np.load(os.path.join(golden_dir, "eigen_probe.npz"))
    cnt = gold["mt_count"]
    o = gold["mt_origin"].reshape(-1, 3)
    d = gold["mt_dir"].reshape(-1, 3)
    tris = gold["mt_tris"].reshape(-1, 3, 3)
    starts = np.concatenate([[0], np.cumsum(cnt)[:-1]])
    checked = hits = box_culled = 0
    for b in range(cnt.size):
        n, k0 = int(cnt[b]), int(starts[b])
        if n > 4:
            continue  # would be split into several leaves: row order is then not traversal order
        c = np.ascontiguousarray(tris[k0:k0 + n][None])
        k = np.full((1, n, 3), 0.5)
        host = ra.HostScene([c], [k])
        ctx.upload(host)
        got = ctx.trace(o[b:b + 1], d[b:b + 1])
        mesh, elem = host.map_prims(got["prim"])
        t = gold["mt_t"][k0:k0 + n]
        row = int(gold["mt_argmin"][b])
        checked += 1
        if np.isfinite(t[row]):
            if elem[0] < 0:
                # the reference tests the leaf's box first (intersect_BLAS) and so does the oracle: a probe
                # whose ray misses the box by rounding is a miss for both
                w = orc.Scene([c], [k]).trace(o[b:b + 1], d[b:b + 1])
                assert w["elem"][0] < 0, b
                box_culled += 1
                continue
            hits += 1
            assert int(elem[0]) == row, b
            assert bits(got["t"])[0] == bits(t[row:row + 1])[0], b
            want_b = np.array([gold["mt_u"][k0 + row], gold["mt_v"][k0 + row], gold["mt_w"][k0 + row]])
            assert np.array_equal(bits(got["bary"][0]), bits(want_b)), b
        else:
            assert elem[0] < 0, b
    assert checked > 500 and hits > 100, (checked, hits, box_culled)


def test_long_sequence_through_the_pipelined_driver(ctx, tmp_path):
    """14 timesteps x 2 cameras through rt_render_scene: more timesteps than staged device buffer sets, so the
    helper-thread uploads, the pointer swaps, the alternating frame buffers and the jitter slots are all reused
    several times while frames are in flight.  Every file must be the oracle's frame of the one continuous
    stream."""
    T = 14
    c, k = meshgen.plate(40, T=T, sway=0.15)
    c2, k2 = meshgen.uv_sphere(14, 18, radius=1.2, center=(0.8, -0.4, -3.0), T=T)
    c2 = c2 + np.linspace(0.0, 0.5, T)[:, None, None, None]  # the sphere drifts
    c3, k3 = meshgen.random_soup(400, seed=3, T=T)
    H, W, spp = 96, 128, 6
    cams = [meshgen.camera_arrays(W, H), meshgen.camera_arrays(W, H, center=(0.6, 0.3, 0.4), target=(0, 0, -2), fov=70)]
    sc = ra.Scene()
    for cc_, kk_ in ((c, k), (c2, k2), (c3, k3)):
        sc.add_mesh(np.ascontiguousarray(cc_), None, None, np.ascontiguousarray(kk_), T)
    for cam in cams:
        sc.add_camera(*cam)
    out = tmp_path / "seq"
    out.mkdir()
    from raytracer_amd import _lib as L
    base = int(L.lib.rt_driver_stream_position(0))  # the driver's stream persists across calls, like the reference's
    ra.render_scene(H, W, sc, spp, out)
    assert len(os.listdir(out)) == 2 * T
    assert int(L.lib.rt_driver_stream_position(0)) == base + 2 * T * H * W * spp * 4
    frame_words = H * W * spp * 4
    hdr = f"P6\n{W} {H}\n255\n".encode()
    meshes = [np.ascontiguousarray(c), np.ascontiguousarray(c2), np.ascontiguousarray(c3)]
    cols = [k, k2, k3]
    g = orc.Rng()
    g.discard(base)
    for t in range(T):
        osc = orc.Scene(meshes, cols, t)
        for ci, cam in enumerate(cams):
            want = osc.render(*cam, H, W, spp, rng=g, want_hits=False)
            assert (out / f"rtimage_{t}_cam{ci}.ppm").read_bytes() == hdr + want["rgb"].tobytes(), (t, ci)


def test_denormal_far_distance_still_counts_as_in_front(ctx):
    """intersect_AABB accepts a box iff tmin <= tmax && tmax > 0 (rtrayintersection.cpp:143).  The packet
    kernel folds both into max(tmin, 2^-1074) <= tmax, which is only the same thing if fp64 denormals are
    neither flushed nor treated as zero: a sheet 1e-320 in front of the camera (tmax denormal) must be hit,
    the same sheet 1e-320 behind it (tmax < 0) must not, exactly as the oracle says."""
    for z, expect_hit in ((-1e-320, True), (1e-320, False), (-0.0, False)):
        tri = np.array([[[[-1.0, -1.0, z], [1.0, -1.0, z], [0.0, 1.0, z]]]])
        col = np.full((1, 1, 3), 0.25)
        host = ra.HostScene([tri], [col])
        ctx.upload(host)
        cam = np.zeros(3)
        p00 = np.array([0.25, 0.125, -1.0])  # every direction component non-zero: the FAST slab path
        sp = np.array([[1e-3, 0.0, 0.0], [0.0, -1e-3, 0.0]])
        jit = np.zeros((1, 2))
        ctx.render(cam, p00, sp, 1, 1, 1, jitter=jit, hit_ids=True, hit_t=True)
        got = ctx.download()
        want = orc.Scene([tri], [col]).render(cam, p00, sp, 1, 1, 1, jitter=jit)
        assert (want["hit_elem"][0] >= 0) == expect_hit, z
        assert (got["hit_prim"][0] >= 0) == expect_hit, z
        assert np.array_equal(bits(got["hit_t"]), bits(want["hit_t"]))
        assert np.array_equal(bits(got["intensity"]), bits(want["intensity"]))


@pytest.mark.parametrize("spp", [1, 64])
def test_rays_grazing_triangle_edges_sure_reject_margin(ctx, spp):
    """Rays through points on triangle edges and within 2^-50 .. 2^-20 (relative) either side of them: the zone
    in which the packet kernel's sure-reject test on the numerators (u + v against det * (1 + 2^-30), u and v
    against -2^-300) must hand the lane over to the exact division path.  spp = 64 makes the 64 lanes of a wave
    the samples of one pixel, so the lanes that are sure rejects sit next to the ones in the margin."""
    c, k = meshgen.plate(12, side=8.0, z0=-4.0, amp=0.06)
    tri = c[0]
    rng = np.random.default_rng(77)
    eps = [0.0] + [s * 2.0 ** -e for e in (52, 50, 45, 40, 33, 31, 30, 29, 27, 24, 20) for s in (1.0, -1.0)]
    pts = []
    for e in range(0, tri.shape[0], 2 if spp == 1 else 9):
        G = tri[e].mean(0)
        for a, b in ((0, 1), (1, 2), (2, 0)):
            P = tri[e, a] + rng.uniform(0.02, 0.98) * (tri[e, b] - tri[e, a])
            pts += [P + x * (G - P) for x in eps]
        pts += [tri[e, v] + x * (G - tri[e, v]) for v in range(3) for x in eps[:7]]  # and the vertices
    pts = np.array(pts)
    if spp == 64:
        pts = pts[: (len(pts) // 64) * 64]
        W, H = 8, len(pts) // 64 // 8
        pts = pts[: W * H * 64]
    else:
        W = 64
        H = len(pts) // W
        pts = pts[: W * H]
    proj = pts * (-4.0 / pts[:, 2:3])  # where the ray from the origin crosses the pixel plane z = -4
    pix = np.arange(len(pts)) // spp
    jit = np.stack([proj[:, 0] + 4.0 - (pix % W), 4.0 - proj[:, 1] - (pix // W)], 1)
    cam = (np.zeros(3), np.array([-4.0, 4.0, -4.0]), np.array([[1.0, 0, 0], [0, -1.0, 0]]))
    got, want, _ = check_image(ctx, [c], [k], cam, H, W, spp, jitter=jit)
    hits = (want["hit_elem"] >= 0).mean()
    assert 0.3 < hits < 1.0  # both sides of the edges are represented


def test_64_spp_instantiation_equals_generic_and_is_skipped_for_deep_trees(ctx, monkeypatch):
    """spp = 64 runs render_packet_kernel<.., P64 = true> (pixel pairs summed together, colours of the second
    pixel parked in the upper part of the wave's stack); RT_NO_PAIR64 forces the generic instantiation, and a tree
    needing more than 32 stack entries must get the generic one by itself.  All three against the oracle."""
    c, k = meshgen.plate(40, amp=0.1)
    cam = meshgen.camera_arrays(37, 21)  # odd sizes: partly empty tiles, inactive pixels in a pair
    a, want, _ = check_image(ctx, [c], [k], cam, 21, 37, 64)
    monkeypatch.setenv("RT_NO_PAIR64", "1")
    b, _, _ = check_image(ctx, [c], [k], cam, 21, 37, 64)
    monkeypatch.delenv("RT_NO_PAIR64")
    assert np.array_equal(bits(a["intensity"]), bits(b["intensity"]))
    assert np.array_equal(a["hit_prim"], b["hit_prim"])
    # deep, left-leaning trees: stack_need == 32 (the deepest the 64-spp instantiation takes: the traversal's
    # entries end right below the parked colours) and 50 (packet kernel, but the generic instantiation)
    for n_t, lo, hi in ((65, 32, 32), (100, 33, 128)):
        tris = []
        for i in range(n_t):
            x = 3.0 ** i * 1e-6
            tris.append([[x, -1.0, -3.0], [x * 1.5, -1.0, -3.0], [x, 1.0, -3.0]])
        cd = np.array(tris)[None]
        kd = np.linspace(0.2, 0.8, n_t)[None, :, None].repeat(3, 2)
        host = ra.HostScene([cd], [kd])
        assert lo <= host.info.stack_need <= hi, host.info.stack_need
        # a narrow view just right of x = 0, where the small triangles (the deep end of the tree) are
        cam = (np.zeros(3), np.array([1e-6, 0.5, -3.0]), np.array([[2e-4, 0, 0], [0, -0.125, 0]]))
        got, want, _ = check_image(ctx, [cd], [kd], cam, 8, 16, 64)
        assert (want["hit_elem"] >= 0).mean() > 0.2


def test_pipelined_driver_equals_unpipelined_at_64_spp(tmp_path):
    """profiles/tools/soak_driver.py in small: the same 8-mesh / 2-camera / 64-spp job through the pipelined driver
    (build-ahead, staged uploads, frame q+1 launched before frame q is read back) and, in a child process, with
    RT_NO_PIPELINE=1 RT_BUILD_AHEAD=1: every P6 file byte-identical."""
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(__file__), "..", "profiles", "tools", "soak_driver.py")
    r = subprocess.run([sys.executable, tool, "5", "512"], capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all 10 files identical" in r.stdout


def test_frame_groups_serve_consecutive_small_frames_and_the_state_catches_up(ctx, monkeypatch):
    """Small frames: one jitter slot holds several consecutive frames (one set of jumps per group) and the device
    generator state lags behind the position while a group is consumed.  Ten consecutive frames through rt_render
    must equal the oracle's over one continuous stream (cpp/rtmathutils.h:15-16 is never reseeded); the state,
    a seek and the jitter test seam in the middle of a group must see the true position; the same frames without
    groups (RT_NO_FRAME_GROUPS) are identical."""
    c, k = meshgen.plate(12)
    host = ra.HostScene([c], [k], 0)
    ctx.upload(host)
    o = orc.Scene([c], [k], 0)
    H, W, spp = 9, 14, 3
    cam = meshgen.camera_arrays(W, H)
    fw = H * W * spp * 4

    def run(n_frames, check_state_at=()):
        ctx.rng_seed(5489)
        frames = []
        for f in range(n_frames):
            ctx.render(*cam, H, W, spp, hit_ids=True)
            frames.append(ctx.download())
            assert ctx.rng_tell() == (f + 1) * fw
            if f in check_state_at:
                # the window of raw words before the position (bits 0..30 of the oldest word are not state)
                dev, want = ctx.rng_state(), ra.mt19937_state_at((f + 1) * fw)
                assert np.array_equal(dev[1:], want[1:]) and (dev[0] >> 31) == (want[0] >> 31)
        return frames

    got = run(10, check_state_at=(0, 3, 4, 9))
    rng = orc.Rng()
    for f in range(10):
        want = o.render(*cam, H, W, spp, rng=rng)
        mesh, elem = host.map_prims(got[f]["hit_prim"])
        assert np.array_equal(elem, want["hit_elem"]) and np.array_equal(bits(got[f]["intensity"]), bits(want["intensity"])), f
    # in the middle of a group: the jitter seam continues the stream where the frames stopped, then frames go on
    ctx.rng_seed(5489)
    for f in range(3):
        ctx.render(*cam, H, W, spp)
    j = ctx.rng_jitter(10, 2)
    rng = orc.Rng()
    rng.discard(3 * fw)
    assert np.array_equal(bits(j), bits(rng.doubles(10) - 0.5))
    ctx.render(*cam, H, W, spp, hit_ids=True)
    want = o.render(*cam, H, W, spp, rng=rng)
    assert np.array_equal(bits(ctx.download()["intensity"]), bits(want["intensity"]))
    # a seek back into the middle of what a group covered
    ctx.rng_seek(2 * fw)
    ctx.render(*cam, H, W, spp)
    assert np.array_equal(bits(ctx.download()["intensity"]), bits(got[2]["intensity"]))
    monkeypatch.setenv("RT_NO_FRAME_GROUPS", "1")
    plain = run(10)
    for f in range(10):
        assert np.array_equal(bits(plain[f]["intensity"]), bits(got[f]["intensity"])), f
    host.close()


def test_learnt_sampler_stream_count_keeps_the_stream(ctx, monkeypatch):
    """Look-ahead fills of very large frames change their number of sampler streams once the context has timed a
    launch (rt_api.hip adapt_streams; new jump polynomials come from a helper thread).  The knobs make a small frame
    count as very large and push the learnt count to its maximum: the frames of a loop must stay the oracle's over
    one continuous stream, before, during and after the switch."""
    import time
    c, k = meshgen.plate(12)
    host = ra.HostScene([c], [k], 0)
    ctx.upload(host)
    o = orc.Scene([c], [k], 0)
    H, W, spp = 64, 96, 16
    cam = meshgen.camera_arrays(W, H)
    fw = H * W * spp * 4
    monkeypatch.setenv("RT_ADAPT_MIN_WORDS", "0")
    monkeypatch.setenv("RT_ADAPT_RATIO", "1e-5")
    monkeypatch.setenv("RT_NO_FRAME_GROUPS", "1")  # (groups of frames per slot are not re-planned)
    ctx.rng_seed(5489)
    rng = orc.Rng()
    streams = []
    for f in range(14):
        ctx.render(*cam, H, W, spp, hit_ids=True)
        got = ctx.download()
        streams.append(ctx.stats()["jitter_streams"])
        want = o.render(*cam, H, W, spp, rng=rng)
        mesh, elem = host.map_prims(got["hit_prim"])
        assert np.array_equal(elem, want["hit_elem"]) and np.array_equal(bits(got["intensity"]), bits(want["intensity"])), f
        assert ctx.rng_tell() == (f + 1) * fw
        if f == 2:
            time.sleep(0.5)  # the helper thread's polynomials are there by now
    assert streams[0] >= 1 and max(streams) > streams[0], streams  # the count did change on the way
    host.close()


@pytest.mark.parametrize("par_min", ["0", "100000"])
def test_scene_uploaded_as_four_spans_by_four_threads(ctx, monkeypatch, par_min):
    """Scenes of RT_UPLOAD_PAR_MIN bytes and more (default 64 MiB) are uploaded as four spans, three of them by helper
    threads on streams of their own; the spans are cut at 4 KiB multiples of the three big arrays laid end to end, so they
    cross array borders and, for small scenes, some are empty.  Forced here for scenes of a few KB to a few hundred KB:
    the frames must stay the oracle's."""
    monkeypatch.setenv("RT_UPLOAD_PAR_MIN", par_min)
    rng = np.random.default_rng(8)
    for n_tris in (1, 7, 40, 333, 2500):
        coords, colors = [], []
        for m in range(int(rng.integers(1, 4))):
            c, k = meshgen.random_soup(n_tris, seed=int(rng.integers(1 << 30)), extent=3.0, size=0.8, z_range=(-7.0, -5.0))
            coords.append(c)
            colors.append(k * (0.3 + 0.2 * m))
        cam = meshgen.camera_arrays(40, 24)
        check_image(ctx, coords, colors, cam, 24, 40, 3)
