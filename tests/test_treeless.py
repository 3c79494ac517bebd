"""Treeless check of hit ids: the closest hit of every ray found by testing ALL triangles with the probe-pinned
Moller-Trumbore row (oracle/rt_oracle.c: orc_brute_force -- no tree, no builder, no walk; rays from the pinned pixel
sample + normalized()).  A hit id that equals this argmin cannot owe anything to either BVH builder or to either
traversal, so these tests pin what tests/test_oracle_pins.py's primitives alone could not: that the walks reach the
triangle the reference's arithmetic says is closest.

The reference can only differ from the brute-force answer by CULLING: it tests a triangle only if the ray passes the
slab test of the triangle's leaf and of every ancestor (rtrayintersection.cpp:161-164, 210).  Every disagreement is
therefore listed and must be explained by a failing box test on that chain (explain_disagreements); anything else fails.

CPU (`not gpu`): the oracle's walk on configs[0] (every timestep, full frame), configs[1] (every pixel, a fifth of the
samples) and 1-3-mesh fuzz scenes.  GPU: the product's kernels on the same, configs[1] with every sample.
"""
import numpy as np
import pytest

import orc
import scenes


def bits(a):
    return np.ascontiguousarray(a, np.float64).view(np.uint64)


def all_tris(coords, t):
    """(triangles (N,3,3) of every mesh at timestep t concatenated, mesh id per triangle, element id per triangle)"""
    tri = np.concatenate([c[t] for c in coords])
    mesh = np.concatenate([np.full(c.shape[1], m, np.int32) for m, c in enumerate(coords)])
    elem = np.concatenate([np.arange(c.shape[1], dtype=np.int32) for c in coords])
    return tri, mesh, elem


def explain_disagreements(bad, cam, dirs, tree_scene, mesh_of, elem_of, first):
    """For rays whose reported hit differs from the brute-force argmin: the brute-force triangle's leaf, or one of its
    ancestors (BLAS or TLAS), must fail the reference's own slab test for this ray.  -> number explained."""
    tl = tree_scene.tlas()
    slots = [tree_scene.slot(s) for s in range(len(np.unique(mesh_of)))]
    slot_of_mesh = {s["mesh"]: i for i, s in enumerate(slots)}
    explained = 0
    for r in bad:
        m, e = int(mesh_of[first[r]]), int(elem_of[first[r]])
        s = slots[slot_of_mesh[m]]
        pos = int(np.nonzero(s["indices"] == e)[0][0])
        # chain of BLAS nodes whose range holds `pos`
        chain, k = [], 0
        while True:
            chain.append(s["boxes"][k])
            if s["left"][k] == -1:
                break
            l = s["left"][k]
            k = l if s["min_idx"][l + 1] > pos else l + 1
        # TLAS nodes on the way to the leaf that WALKS this slot (quirk 1: the slot index, not the mesh)
        si = slot_of_mesh[m]
        k = 0
        while True:
            chain.append(tl["boxes"][k])
            if tl["left"][k] == -1:
                break
            l = tl["left"][k]
            k = l if tl["min_blas_idx"][l + 1] > si else l + 1
        d = np.ascontiguousarray(dirs[r])
        o = np.ascontiguousarray(cam, np.float64)
        if any(not orc.lib.orc_slab(o, d, np.ascontiguousarray(b[:3]), np.ascontiguousarray(b[3:])) for b in chain):
            explained += 1
    return explained


def check_against_brute_force(coords, t, cam, W, spp, jitter, hit_mesh, hit_elem, hit_t, tree_scene, sample=None):
    """hit_* are per-sample arrays of one frame.  sample: optional index array of the samples to check."""
    tri, mesh_of, elem_of = all_tris(coords, t)
    dirs = orc.ray_dirs(*cam, W, spp, jitter)
    if sample is not None:
        dirs, hit_mesh, hit_elem, hit_t = dirs[sample], hit_mesh[sample], hit_elem[sample], hit_t[sample]
    bt, first, nmin = orc.brute_force(cam[0], dirs, tri)
    hit = first >= 0
    bm = np.where(hit, mesh_of[np.maximum(first, 0)], -1)
    be = np.where(hit, elem_of[np.maximum(first, 0)], -1)
    same = (bm == hit_mesh) & (be == hit_elem) & (bits(bt) == bits(hit_t))
    # several triangles at exactly the minimal t: any of them is a correct closest hit for THIS check (which one the
    # reference keeps is a matter of visit order, pinned by tests/test_tree_fixtures.py); t must still be the minimum
    tied = (~same) & (nmin > 1) & (bits(bt) == bits(hit_t))
    bad = np.nonzero(~same & ~tied)[0]
    n_expl = explain_disagreements(bad, cam[0], dirs, tree_scene, mesh_of, elem_of, first) if bad.size else 0
    assert n_expl == bad.size, (f"{bad.size - n_expl} of {dirs.shape[0]} rays report another hit than the closest "
                               f"triangle although no box on its chain culls it; first: {bad[:5]}")
    return dict(rays=dirs.shape[0], hits=int(hit.sum()), tied=int(tied.sum()), culled=int(bad.size))


def fuzz_scene(seed):
    """1-3 meshes: a wavy plate, a sphere in front of it, a triangle soup; no TLAS quirk possible below 4 meshes."""
    from raytracer_amd import meshgen
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 4))
    parts = [meshgen.plate(int(rng.integers(6, 30)), side=float(rng.uniform(4, 9)), z0=float(rng.uniform(-5, -3))),
             meshgen.uv_sphere(int(rng.integers(5, 12)), int(rng.integers(6, 14)), radius=float(rng.uniform(0.4, 1.2)),
                               center=(float(rng.uniform(-1, 1)), float(rng.uniform(-1, 1)), -2.5)),
             meshgen.random_soup(int(rng.integers(50, 400)), seed=seed)]
    order = rng.permutation(3)[:n]
    coords = [np.ascontiguousarray(parts[i][0]) for i in order]
    colors = [np.ascontiguousarray(parts[i][1]) for i in order]
    cam = meshgen.camera_arrays(96, 64, center=(float(rng.uniform(-0.3, 0.3)), float(rng.uniform(-0.3, 0.3)), 0.0),
                                target=(0.0, 0.0, -3.0), fov=float(rng.uniform(50, 95)))
    return coords, colors, cam, 64, 96, int(rng.integers(1, 6))


# ------------------------------------------------------------------------------------------------ CPU: the oracle
def oracle_frame(coords, colors, t, cam, H, W, spp, jitter):
    sc = orc.Scene(coords, colors, t)
    w = sc.render(*cam, H, W, spp, jitter=jitter)
    return sc, w


def test_oracle_block_every_timestep_full_frame():
    coords, colors, cams, spp = scenes.block()
    rng = orc.Rng()
    for t in range(coords[0].shape[0]):
        jit = orc.jitter_stream(scenes.H * scenes.W * spp, rng)  # one continuous stream, like render_scene
        sc, w = oracle_frame(coords, colors, t, cams[0], scenes.H, scenes.W, spp, jit)
        r = check_against_brute_force(coords, t, cams[0], scenes.W, spp, jit, w["hit_mesh"], w["hit_elem"], w["hit_t"], sc)
        assert r["hits"] > 0.05 * r["rays"] and r["culled"] == 0, (t, r)


def test_oracle_sphere_every_pixel():
    coords, colors, cams, spp = scenes.sphere()
    jit = orc.jitter_stream(scenes.H * scenes.W * spp)
    sc, w = oracle_frame(coords, colors, 0, cams[0], scenes.H, scenes.W, spp, jit)
    sample = np.arange(0, scenes.H * scenes.W * spp, 5)  # 50 spp: every pixel keeps samples 0, 5, .. 45
    r = check_against_brute_force(coords, 0, cams[0], scenes.W, spp, jit, w["hit_mesh"], w["hit_elem"], w["hit_t"], sc,
                                  sample)
    assert r["hits"] > 0.1 * r["rays"] and r["culled"] == 0, r


@pytest.mark.parametrize("seed", range(6))
def test_oracle_fuzz_scenes(seed):
    coords, colors, cam, H, W, spp = fuzz_scene(seed)
    jit = orc.jitter_stream(H * W * spp)
    sc, w = oracle_frame(coords, colors, 0, cam, H, W, spp, jit)
    r = check_against_brute_force(coords, 0, cam, W, spp, jit, w["hit_mesh"], w["hit_elem"], w["hit_t"], sc)
    assert r["hits"] > 0.05 * r["rays"], r


# ------------------------------------------------------------------------------------------------ GPU: the product
def gpu_frame(ctx, coords, colors, t, cam, H, W, spp, pos):
    import raytracer_amd as ra
    host = ra.HostScene(coords, colors, t)
    ctx.upload(host)
    ctx.rng_seed(5489)
    ctx.rng_seek(pos)
    ctx.render(*cam, H, W, spp, hit_ids=True, hit_t=True)
    g = ctx.download(pinned=False)
    mesh, elem = host.map_prims(g["hit_prim"])
    return host, mesh, elem, g["hit_t"]


@pytest.fixture
def ctx():
    import raytracer_amd as ra
    c = ra.Context(0)
    yield c
    c.close()


@pytest.mark.gpu
def test_gpu_block_every_timestep_full_frame(ctx):
    coords, colors, cams, spp = scenes.block()
    rng = orc.Rng()
    words = scenes.H * scenes.W * spp * 4
    for t in range(coords[0].shape[0]):
        jit = orc.jitter_stream(scenes.H * scenes.W * spp, rng)
        host, mesh, elem, ht = gpu_frame(ctx, coords, colors, t, cams[0], scenes.H, scenes.W, spp, t * words)
        r = check_against_brute_force(coords, t, cams[0], scenes.W, spp, jit, mesh, elem, ht, host)
        assert r["hits"] > 0.05 * r["rays"] and r["culled"] == 0, (t, r)


@pytest.mark.gpu
def test_gpu_sphere_every_sample(ctx):
    coords, colors, cams, spp = scenes.sphere()
    jit = orc.jitter_stream(scenes.H * scenes.W * spp)
    host, mesh, elem, ht = gpu_frame(ctx, coords, colors, 0, cams[0], scenes.H, scenes.W, spp, 0)
    r = check_against_brute_force(coords, 0, cams[0], scenes.W, spp, jit, mesh, elem, ht, host)
    assert r["rays"] == 4_500_000 and r["hits"] > 0.1 * r["rays"] and r["culled"] == 0, r


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(6))
def test_gpu_fuzz_scenes(ctx, seed):
    coords, colors, cam, H, W, spp = fuzz_scene(seed)
    jit = orc.jitter_stream(H * W * spp)
    host, mesh, elem, ht = gpu_frame(ctx, coords, colors, 0, cam, H, W, spp, 0)
    r = check_against_brute_force(coords, 0, cam, W, spp, jit, mesh, elem, ht, host)
    assert r["hits"] > 0.05 * r["rays"], r


# ---- the culling explanation itself, exercised: with >= 4 meshes the reference's TLAS leaves may guard the wrong
# meshes (quirk 1), so rays DO miss their geometrically closest triangle -- and every such ray must be explained by a
# failing box on the chain to the leaf that walks the triangle's slot
def quirk_frame(render):
    coords, colors, cams, _ = scenes.stereo(T=1)
    H, W, spp = 54, 96, 3
    cam = cams[0]
    from raytracer_amd import meshgen
    cam = meshgen.camera_arrays(W, H, center=(-0.4, 0.0, 0.0), target=(-1.0, -1.0, -6.0), fov=75)
    jit = orc.jitter_stream(H * W * spp)
    sc, mesh, elem, ht = render(coords, colors, cam, H, W, spp, jit)
    r = check_against_brute_force(coords, 0, cam, W, spp, jit, mesh, elem, ht, sc)
    assert r["culled"] > 100 and r["hits"] > r["culled"], r


def test_oracle_quirk1_disagreements_are_all_explained_by_a_culling_box():
    def render(coords, colors, cam, H, W, spp, jit):
        sc, w = oracle_frame(coords, colors, 0, cam, H, W, spp, jit)
        return sc, w["hit_mesh"], w["hit_elem"], w["hit_t"]
    quirk_frame(render)


@pytest.mark.gpu
def test_gpu_quirk1_disagreements_are_all_explained_by_a_culling_box(ctx):
    def render(coords, colors, cam, H, W, spp, jit):
        return gpu_frame(ctx, coords, colors, 0, cam, H, W, spp, 0)
    quirk_frame(render)
