"""Cuts (DESIGN.md 4.2 "cuts"; raytracer_amd/csrc/rt_kernels.hip.hpp cut_refine) checked on the CPU.

The render kernel starts the walk of a block's rays at the block's cut instead of at the TLAS root, on the
strength of an argument (BLAS monotonicity + a conservative classification of boxes against the block's ray
pyramid).  Here the host twin of the cut kernel (rt_tile_cuts_host: the same code compiled for the host) is
checked against a brute-force numpy walk with the reference's own slab test (cpp/rtrayintersection.cpp:124-144):
for rays through every block -- its extreme sample positions and random ones -- the leaves reached from the cut's
entries, in order, must be exactly the leaves the reference's walk from node 0 reaches
(rtrayintersection.cpp:146-228), and the header's promises about the rays' direction signs must hold.
"""
import numpy as np
import pytest

import raytracer_amd as ra
from raytracer_amd import meshgen

LEAF, TLAS, ROOT = 0x80000000, 0x40000000, 0x20000000
NODE = TLAS | ROOT
TRIM, NODE_IDX = 0x10000000, 0x03FFFFFF
SIMPLE = 0x00010000  # header: the list holds nothing but leaves (<= 7 triangles) and TRIM pairs
N_SIMPLE = [0]      # blocks with that promise seen by check_scene


class Tree:
    """The host tree in the reference's numbering plus what maps device records back to it."""

    def __init__(self, host, coords, timestep):
        self.tl = host.tlas()
        self.slots = [host.slot(s) for s in range(host.info.n_meshes)]
        self.roots = np.array([s["dev_root"] for s in self.slots])
        pm, pe = host.prim_tables()
        self.verts = np.stack([np.asarray(coords[m], np.float64)[timestep, e] for m, e in zip(pm, pe)])  # prim -> 3 x 3

    def locate(self, dev_idx):
        s = int(np.searchsorted(self.roots, dev_idx, side="right") - 1)
        i = int(dev_idx - self.roots[s])
        assert 0 <= i < len(self.slots[s]["left"]), (dev_idx, s, i)
        return s, i

    def leaf_range(self, s, i):
        sl = self.slots[s]
        return int(sl["tri_base"] + sl["min_idx"][i]), int(sl["count"][i])


def make_rays(cam, p00, sp, x0, x1, y0, y1, rng, n_random=6):
    """Sample positions of the pixels [x0,x1] x [y0,y1]: the four extreme corners (offsets -0.5 / +0.5) and random
    ones; directions exactly as cpp/rtrender.cpp:53-58 computes them."""
    sp = np.asarray(sp, np.float64).reshape(2, 3)
    s0 = np.concatenate([[x0 - 0.5, x1 + 0.5, x0 - 0.5, x1 + 0.5], rng.uniform(x0 - 0.5, x1 + 0.5, n_random)])
    s1 = np.concatenate([[y0 - 0.5, y0 - 0.5, y1 + 0.5, y1 + 0.5], rng.uniform(y0 - 0.5, y1 + 0.5, n_random)])
    dd = ((p00[None, :] + s0[:, None] * sp[0][None, :]) + s1[:, None] * sp[1][None, :]) - cam[None, :]
    z = (dd[:, 0] * dd[:, 0] + dd[:, 1] * dd[:, 1]) + dd[:, 2] * dd[:, 2]
    d = dd / np.sqrt(z)[:, None]
    with np.errstate(divide="ignore"):
        inv = 1.0 / d
    return d, inv


def slab(box, cam, inv):
    """intersect_AABB for all rays (no NaN operand here, so min/max are plain)."""
    with np.errstate(invalid="ignore", over="ignore"):
        t0 = (box[None, :3] - cam[None, :]) * inv
        t1 = (box[None, 3:] - cam[None, :]) * inv
        lo, hi = np.minimum(t0, t1), np.maximum(t0, t1)
        tmin = np.maximum(np.maximum(lo[:, 0], lo[:, 1]), lo[:, 2])
        tmax = np.minimum(np.minimum(hi[:, 0], hi[:, 1]), hi[:, 2])
        return (tmin <= tmax) & (tmax > 0.0) & (tmin < np.inf)


def tri_hits(v, cam, d):
    """One row of intersect_bvh_triangles (cpp/rtrayintersection.cpp:64-106) for all rays: hit or not."""
    v0, e0, e2 = v[0], v[1] - v[0], v[2] - v[0]
    px = d[:, 1] * e2[2] - d[:, 2] * e2[1]
    py = d[:, 2] * e2[0] - d[:, 0] * e2[2]
    pz = d[:, 0] * e2[1] - d[:, 1] * e2[0]
    det = (e0[0] * px + e0[1] * py) + e0[2] * pz
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        inv = 1.0 / det
        tv = cam - v0
        u = ((tv[0] * px + tv[1] * py) + tv[2] * pz) * inv
        q = np.array([tv[1] * e0[2] - tv[2] * e0[1], tv[2] * e0[0] - tv[0] * e0[2], tv[0] * e0[1] - tv[1] * e0[0]])
        vv = ((d[:, 0] * q[0] + d[:, 1] * q[1]) + d[:, 2] * q[2]) * inv
        t = ((e2[0] * q[0] + e2[1] * q[1]) + e2[2] * q[2]) * inv
        return (det > 1e-6) & (u >= 0) & (u <= 1) & (vv >= 0) & (u + vv <= 1) & (t >= 0)


def walk_blas(tree, s, i, mask, cam, inv, out, own_test=True):
    sl = tree.slots[s]
    if own_test:
        mask = mask & slab(sl["boxes"][i], cam, inv)
    if not mask.any():
        return
    if sl["left"][i] < 0:
        out.append((*tree.leaf_range(s, i), mask.copy()))
        return
    walk_blas(tree, s, int(sl["left"][i]), mask, cam, inv, out)
    walk_blas(tree, s, int(sl["left"][i]) + 1, mask, cam, inv, out)


def walk_tlas(tree, i, mask, cam, inv, out):
    tl = tree.tl
    mask = mask & slab(tl["boxes"][i], cam, inv)
    if not mask.any():
        return
    if tl["left"][i] < 0:
        for s in range(int(tl["min_blas_idx"][i]), int(tl["min_blas_idx"][i] + tl["blas_count"][i])):
            walk_blas(tree, s, 0, mask, cam, inv, out)
    else:
        walk_tlas(tree, int(tl["left"][i]), mask, cam, inv, out)
        walk_tlas(tree, int(tl["left"][i]) + 1, mask, cam, inv, out)


def walk_cut(tree, entries, n_rays, cam, inv):
    """-> [(first triangle, count, rays)] in the order the render kernel tests them."""
    out = []
    everyone = np.ones(n_rays, bool)
    entries = [int(e) for e in entries]
    k = 0
    while k < len(entries):
        e = entries[k]
        k += 1
        if e & LEAF:  # a leaf every ray passes, or a run of its live triangles (cut_trim)
            assert ((e >> 28) & 7) != 7, "leaf_ext leaves are not used by these scenes"
            out.append((e & 0x0FFFFFFF, ((e >> 28) & 7) + 1, everyone.copy()))
        elif (e & NODE) == NODE:
            s, i = tree.locate(e & NODE_IDX)
            if e & TRIM:  # a leaf tested per ray; the next word names its live triangles
                r = entries[k]
                k += 1
                assert tree.slots[s]["left"][i] < 0 and (r & LEAF) and ((r >> 28) & 7) != 7
                m = slab(tree.slots[s]["boxes"][i], cam, inv)
                f, c = tree.leaf_range(s, i)
                assert f <= (r & 0x0FFFFFFF) and (r & 0x0FFFFFFF) + ((r >> 28) & 7) + 1 <= f + c
                if m.any():
                    out.append((r & 0x0FFFFFFF, ((r >> 28) & 7) + 1, m))
            else:
                walk_blas(tree, s, i, everyone, cam, inv, out, own_test=True)
        elif e & TLAS:
            walk_tlas(tree, e & 0x3FFFFFFF, everyone, cam, inv, out)
        elif e & ROOT:
            walk_blas(tree, e & 0x1FFFFFFF, 0, everyone, cam, inv, out)
        else:  # internal BLAS node every ray passes: e = device index of its left child
            s, i = tree.locate(e)
            walk_blas(tree, s, i, everyone, cam, inv, out)
            walk_blas(tree, s, i + 1, everyone, cam, inv, out)
    return out


def per_ray(seq, n_rays):
    return [[t for f, c, m in seq if m[r] for t in range(f, f + c)] for r in range(n_rays)]


def check_scene(coords, colors, cam, H, W, spp, level, seed, timestep=0):
    """-> blocks with a cut, entries, blocks, triangles the reference tests, triangles the cut walk tests."""
    host = ra.HostScene(coords, colors, timestep)
    tree = Tree(host, coords, timestep)
    cam_c, p00, sp = (np.asarray(x, np.float64) for x in cam)
    words, tx, (lo, top) = ra.tile_cuts_host(host, cam_c, p00, sp, H, W, spp, level=level)
    assert lo <= level <= top
    ts = 1 << level
    rng = np.random.default_rng(seed)
    n_valid = n_entries = n_ref = n_cut = 0
    for b in range(words.shape[0]):
        hdr = int(words[b, 0])
        if not hdr >> 31:
            continue
        n_valid += 1
        n = hdr & 0xff
        n_entries += n
        bx, by = b % tx, b // tx
        x0, y0 = bx * ts, by * ts
        x1, y1 = min(x0 + ts - 1, W - 1), min(y0 + ts - 1, H - 1)
        d, inv = make_rays(cam_c, p00, sp.reshape(6), x0, x1, y0, y1, rng)
        # the header's promise: on an axis of kind 0 / 1 every ray of the block has that sign, and is tame
        for a in range(3):
            kind = (hdr >> (8 + 2 * a)) & 3
            if kind == 0:
                assert (inv[:, a] > 0).all() and (inv[:, a] < 1e100).all(), (b, a)
            elif kind == 1:
                assert (inv[:, a] < 0).all() and (inv[:, a] > -1e100).all(), (b, a)
        if hdr & SIMPLE:  # the render kernel then runs a loop that knows only these two kinds of entries
            N_SIMPLE[0] += 1
            ents = [int(e) for e in words[b, 1:1 + n]]
            k = 0
            while k < len(ents):
                e = ents[k]
                if e & LEAF:
                    assert ((e >> 28) & 7) != 7, (b, hex(e))
                    k += 1
                else:
                    assert (e & (NODE | TRIM)) == (NODE | TRIM) and k + 1 < len(ents), (b, hex(e))
                    r = ents[k + 1]
                    assert (r & LEAF) and ((r >> 28) & 7) != 7, (b, hex(r))
                    k += 2
        tame = (np.abs(inv) < 1e100).all(axis=1)  # the kernel walks batches with other rays from the root
        ref = []
        walk_tlas(tree, 0, tame.copy(), cam_c, inv, ref)
        got = walk_cut(tree, words[b, 1:1 + n], len(tame), cam_c, inv)
        cut = [(f, c, m & tame) for f, c, m in got]
        want, have = per_ray(ref, len(tame)), per_ray(cut, len(tame))
        for r in range(len(tame)):
            # the cut walk tests the triangles the reference tests, in its order, but for some that this ray misses
            # by the reference's own arithmetic
            it = iter(want[r])
            dropped = []
            for t in have[r]:
                for w in it:
                    if w == t:
                        break
                    dropped.append(w)
                else:
                    pytest.fail(f"block {b} ray {r}: triangle {t} out of order / not in the reference's walk: "
                                f"{[hex(int(x)) for x in words[b, :1 + n]]}")
            dropped.extend(it)
            for t in dropped:
                assert not tri_hits(tree.verts[t], cam_c, d[r:r + 1])[0], (b, r, t)
            n_ref += len(want[r])
            n_cut += len(have[r])
    host.close()
    return n_valid, n_entries, words.shape[0], n_ref, n_cut


def two_meshes():
    c0, k0 = meshgen.plate(10, T=2, sway=0.15)
    c1, k1 = meshgen.uv_sphere(6, 8, radius=0.8, center=(0.4, 0.3, -2.5), T=2)
    return [c0, c1], [k0, k1]


@pytest.mark.parametrize("level", [1, 2, 3])
def test_cut_walk_equals_root_walk_two_meshes_rotated_camera(level, monkeypatch):
    if level == 2:
        monkeypatch.setenv("RT_CUT_STEP", "1")  # every level is made (default: the top, every second one, the finest)
    coords, colors = two_meshes()
    H, W = 20, 32
    for ci, cam in enumerate([meshgen.camera_arrays(W, H),
                              meshgen.camera_arrays(W, H, center=(0.3, 0.1, 0.2), target=(0, 0, -2), fov=70),
                              meshgen.camera_arrays(W, H, center=(3.0, 2.0, 1.0), target=(0.4, 0.3, -2.5), fov=50)]):
        n_valid, n_entries, n_blocks, _, _ = check_scene(coords, colors, cam, H, W, 64, level, seed=ci, timestep=1)
        assert n_valid == n_blocks  # mixed-sign blocks get cuts too


def test_cut_walk_pixel_level_and_coarse_level(monkeypatch):
    """RT_CUT_LO=0 makes per-pixel cuts; spp = 1 asks for nothing finer than 8 x 8."""
    coords, colors = two_meshes()
    H, W = 16, 24
    cam = meshgen.camera_arrays(W, H, center=(0.3, 0.1, 0.2), target=(0, 0, -2), fov=70)
    monkeypatch.setenv("RT_CUT_LO", "0")
    check_scene(coords, colors, cam, H, W, 64, 0, seed=5)
    monkeypatch.delenv("RT_CUT_LO")
    host = ra.HostScene(coords, colors, 0)
    _, _, (lo, top) = ra.tile_cuts_host(host, *cam, H, W, 1)
    assert lo == 3 and top >= lo
    host.close()
    check_scene(coords, colors, cam, H, W, 1, 3, seed=6)


def test_cut_walk_eight_meshes_with_the_slot_quirk():
    """8 meshes placed so that the TLAS stores its BLASes in another order than it indexes them (quirk 1,
    rtbvh.cpp:552-560 vs rtrayintersection.cpp:214-218): TLAS tests are kept unless the whole block is sure."""
    def slot_order(host):
        tl = host.tlas()
        return [int(tl["min_blas_idx"][i]) for i in range(len(tl["left"])) if tl["left"][i] < 0]

    for seed in range(40):  # the first placement for which the quirk is active (deterministic)
        rng = np.random.default_rng(seed)
        coords, colors = [], []
        for m in range(8):
            c, k = meshgen.plate(6, side=2.0, x0=float(rng.uniform(-5, 3)), y0=float(rng.uniform(-4, 2)),
                                 z0=float(rng.uniform(-9, -4)))
            coords.append(c)
            colors.append(k)
        host = ra.HostScene(coords, colors, 0)
        order = slot_order(host)
        host.close()
        if order != sorted(order):
            break
    else:
        pytest.fail("no placement triggers the quirk")
    H, W = 24, 32
    for level in (1, 3):
        check_scene(coords, colors, meshgen.camera_arrays(W, H, fov=80), H, W, 64, level, seed=level)


def test_cut_walk_camera_inside_the_cloud_and_far_from_the_origin():
    """Boxes around and behind the camera; a camera 1e4 units from the origin (the rounding of the sample position
    enters the margins); back faces are irrelevant here (only boxes are classified)."""
    c, k = meshgen.random_soup(600, seed=21, extent=4.0, z_range=(-4.0, 4.0))
    H, W = 16, 16
    cams = [meshgen.camera_arrays(W, H, center=(0.2, -0.1, 0.3), target=(1.0, 0.5, -1.0), fov=100),
            meshgen.camera_arrays(W, H, center=(0.0, 0.0, 0.0), target=(0.2, 1.0, -0.3), fov=60)]
    for ci, cam in enumerate(cams):
        check_scene([c], [k], cam, H, W, 64, 1, seed=30 + ci)
    off = np.array([1.0e4, -2.0e4, 5.0e3])
    c2 = c + off[None, None, None, :]
    cam = meshgen.camera_arrays(W, H, center=tuple(off + [0.2, -0.1, 6.0]), target=tuple(off), fov=70)
    check_scene([c2], [k], cam, H, W, 64, 1, seed=33)


def test_cuts_prune_most_of_the_tree():
    """Sanity of the point of it all: on a dense plate a 2 x 2-pixel block's cut is a handful of entries."""
    c, k = meshgen.plate(40)
    H = W = 96
    host = ra.HostScene([c], [k], 0)
    words, tx, (lo, top) = ra.tile_cuts_host(host, *meshgen.camera_arrays(W, H), H, W, 64)
    assert lo == 1
    n = words[:, 0] & 0xff
    assert (words[:, 0] >> 31).all() and n.mean() < 6 and host.info.n_blas_nodes > 1500
    host.close()


def test_trimmed_leaves_drop_triangles_and_only_sure_misses(monkeypatch):
    """cut_trim: at the finest level the lists name the leaves' live triangles.  check_scene asserts that every
    triangle the cut walk skips is one the ray misses by the reference's own arithmetic; here also that trimming
    does something, and that RT_NO_TRIM=1 gives the untrimmed lists (every triangle of the reference's walk)."""
    c, k = meshgen.plate(40, sway=0.2)
    c1, k1 = meshgen.uv_sphere(12, 16, radius=0.8, center=(0.2, 0.1, -1.5))  # closed surface: back faces
    H = W = 48
    cam = meshgen.camera_arrays(W, H, center=(0.1, 0.05, 0.4), target=(0.0, 0.0, -2.0), fov=60)
    N_SIMPLE[0] = 0
    n_valid, _, _, n_ref, n_cut = check_scene([c, c1], [k, k1], cam, H, W, 64, 1, seed=11)
    assert n_cut < 0.7 * n_ref, (n_ref, n_cut)
    assert N_SIMPLE[0] > 0.5 * n_valid, (N_SIMPLE[0], n_valid)  # most finest-level lists are leaves and pairs only
    monkeypatch.setenv("RT_NO_TRIM", "1")
    N_SIMPLE[0] = 0
    _, _, _, n_ref2, n_cut2 = check_scene([c, c1], [k, k1], cam, H, W, 64, 1, seed=11)
    assert n_ref2 == n_ref and n_cut2 == n_ref and N_SIMPLE[0] == 0
    monkeypatch.delenv("RT_NO_TRIM")
    monkeypatch.setenv("RT_CUT_LO", "0")  # per-pixel blocks
    _, _, _, n_ref0, n_cut0 = check_scene([c, c1], [k, k1], cam, H, W, 64, 0, seed=12)
    assert n_cut0 < 0.6 * n_ref0, (n_ref0, n_cut0)
