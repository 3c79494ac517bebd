"""Full-size checks on the GPU at BASELINE-scale inputs, through properties that do not need the oracle to
trace every ray (it does ~1 Mray/s): stratified oracle rows at exact RNG positions, idempotence, row-band
decomposition, packet kernel == per-lane kernel, and self-consistency of every reported hit (the hit triangle,
re-intersected in numpy with the reference's formula order, reproduces the reported t bit for bit)."""
import os

import numpy as np
import pytest

import orc
import raytracer_amd as ra
from raytracer_amd import meshgen

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float64).view(np.uint64)


@pytest.fixture(scope="module")
def big():
    """>= 1M-triangle deformed plate (north_star's target size), 1024 x 768, 8 spp = 6.3 M rays."""
    c, k = meshgen.plate(724)
    H, W, spp = 768, 1024, 8
    cam = meshgen.camera_arrays(W, H)
    host = ra.HostScene([c], [k])
    ctx = ra.Context(0)
    ctx.upload(host)
    ctx.rng_seed(5489)
    ctx.render(*cam, H, W, spp, hit_ids=True, hit_t=True)
    out = ctx.download()
    yield dict(c=c, k=k, H=H, W=W, spp=spp, cam=cam, host=host, ctx=ctx, out=out)
    ctx.close()


def test_oracle_rows_at_exact_stream_positions(big):
    H, W, spp, cam = big["H"], big["W"], big["spp"], big["cam"]
    o = orc.Scene([big["c"]], [big["k"]])
    for r in (0, 97, 383, 384, 767):
        rng = orc.Rng()
        rng.set_state(ra.mt19937_state_at(r * W * spp * 4))
        w = o.render(*cam, H, W, spp, rng=rng, row0=r, row1=r + 1)
        sl = slice(r * W * spp, (r + 1) * W * spp)
        mesh, elem = big["host"].map_prims(big["out"]["hit_prim"][sl])
        assert np.array_equal(elem, w["hit_elem"]) and np.array_equal(mesh, w["hit_mesh"]), r
        assert np.array_equal(bits(big["out"]["hit_t"][sl]), bits(w["hit_t"])), r
        assert np.array_equal(bits(big["out"]["intensity"][r]), bits(w["intensity"][0])), r
        assert np.array_equal(big["out"]["rgb"][r], w["rgb"][0]), r


def test_hit_ids_against_the_treeless_argmin(big):
    """Builder- and walk-independent: 24 stretches of 64 pixels x 8 samples, closest of ALL 1,048,352 triangles."""
    H, W, spp, cam, out = big["H"], big["W"], big["spp"], big["cam"], big["out"]
    rs = np.random.default_rng(7)
    n_hit = 0
    for j, i0 in zip(rs.integers(0, H, 24), rs.integers(0, W - 64, 24)):
        s0 = (int(j) * W + int(i0)) * spp
        rng = orc.Rng()
        rng.set_state(ra.mt19937_state_at(s0 * 4))
        dirs = orc.ray_dirs_segment(*cam, spp, int(j), int(i0), 64, rng.doubles(64 * spp * 2) - 0.5)
        bt, first, nmin = orc.brute_force(cam[0], dirs, big["c"][0])
        _, elem = big["host"].map_prims(out["hit_prim"][s0:s0 + 64 * spp])
        ok = (elem == first) | ((nmin > 1) & (elem >= 0))
        assert ok.all() and np.array_equal(bits(bt), bits(out["hit_t"][s0:s0 + 64 * spp])), (j, i0)
        n_hit += int((first >= 0).sum())
    assert n_hit > 0.5 * 24 * 64 * spp


def test_idempotent_and_band_decomposition(big):
    H, W, spp, cam, ctx = big["H"], big["W"], big["spp"], big["cam"], big["ctx"]
    ctx.rng_seed(5489)
    ctx.render(*cam, H, W, spp, hit_ids=True)
    again = ctx.download()
    assert np.array_equal(again["rgb"], big["out"]["rgb"])
    assert np.array_equal(again["hit_prim"], big["out"]["hit_prim"])
    assert np.array_equal(bits(again["intensity"]), bits(big["out"]["intensity"]))
    parts = []
    for a, b in [(0, 250), (250, 251), (251, 768)]:
        ctx.rng_seek(a * W * spp * 4)
        ctx.render(*cam, H, W, spp, rows=(a, b), hit_ids=True)
        parts.append(ctx.download())
    assert np.array_equal(np.concatenate([p["hit_prim"] for p in parts]), big["out"]["hit_prim"])
    assert np.array_equal(bits(np.concatenate([p["intensity"] for p in parts])), bits(big["out"]["intensity"]))


def test_packet_kernel_equals_per_lane_kernel(big):
    H, W, spp, cam, ctx = big["H"], big["W"], big["spp"], big["cam"], big["ctx"]
    os.environ["RT_FORCE_PERLANE"] = "1"
    try:
        ctx.rng_seed(5489)
        ctx.render(*cam, H, W, spp, hit_ids=True, hit_t=True)
        per_lane = ctx.download()
    finally:
        del os.environ["RT_FORCE_PERLANE"]
    assert np.array_equal(per_lane["hit_prim"], big["out"]["hit_prim"])
    assert np.array_equal(bits(per_lane["hit_t"]), bits(big["out"]["hit_t"]))
    assert np.array_equal(bits(per_lane["intensity"]), bits(big["out"]["intensity"]))
    assert np.array_equal(per_lane["rgb"], big["out"]["rgb"])


def test_every_hit_is_self_consistent(big):
    """Recompute t for each reported (ray, triangle) pair in numpy with the reference's operation order
    (rtrayintersection.cpp:64-106); the jitter comes from the oracle's generator for whole rows."""
    H, W, spp, cam = big["H"], big["W"], big["spp"], big["cam"]
    cc, p00, sp = cam
    pm, pe = big["host"].prim_tables()
    tris = big["c"][0]
    rows = np.array([5, 300, 700])
    n_checked = 0
    for r in rows:
        rng = orc.Rng()
        rng.set_state(ra.mt19937_state_at(int(r) * W * spp * 4))
        jit = rng.doubles(W * spp * 2).reshape(W, spp, 2) - 0.5
        i = np.arange(W, dtype=np.float64)[:, None]
        s0 = i + jit[..., 0]
        s1 = float(r) + jit[..., 1]
        dd = np.stack([(p00[c] + s0 * sp[0][c]) + s1 * sp[1][c] - cc[c] for c in range(3)], -1)
        z = (dd[..., 0] * dd[..., 0] + dd[..., 1] * dd[..., 1]) + dd[..., 2] * dd[..., 2]
        d = dd / np.sqrt(z)[..., None]
        sl = slice(int(r) * W * spp, (int(r) + 1) * W * spp)
        prim = big["out"]["hit_prim"][sl].reshape(W, spp)
        t_gpu = big["out"]["hit_t"][sl].reshape(W, spp)
        hit = prim >= 0
        n_checked += int(hit.sum())
        tri = tris[pe[prim[hit]]]
        dv = d[hit]
        e0 = tri[:, 1] - tri[:, 0]
        e2 = tri[:, 2] - tri[:, 0]
        px = dv[:, 1] * e2[:, 2] - dv[:, 2] * e2[:, 1]
        py = dv[:, 2] * e2[:, 0] - dv[:, 0] * e2[:, 2]
        pz = dv[:, 0] * e2[:, 1] - dv[:, 1] * e2[:, 0]
        det = (e0[:, 0] * px + e0[:, 1] * py) + e0[:, 2] * pz
        inv = 1.0 / det
        tv = cc[None, :] - tri[:, 0]
        qx = tv[:, 1] * e0[:, 2] - tv[:, 2] * e0[:, 1]
        qy = tv[:, 2] * e0[:, 0] - tv[:, 0] * e0[:, 2]
        qz = tv[:, 0] * e0[:, 1] - tv[:, 1] * e0[:, 0]
        t = ((e2[:, 0] * qx + e2[:, 1] * qy) + e2[:, 2] * qz) * inv
        assert np.array_equal(bits(t), bits(t_gpu[hit]))
        assert np.all(np.isinf(t_gpu[~hit]))
    assert n_checked > 10000


def test_gpu_visit_statistics_equal_the_surveys_probe_of_the_compiled_reference():
    """GPU twin of tests/test_oracle_pins.py's last test, through RT_FLAG_COUNTERS: product builder + packet
    kernel reproduce the node / triangle visits per ray and the hit rate the survey measured on the compiled
    reference (43.9 / 4.39 / 0.906) -- and equal the oracle's counts exactly."""
    from test_oracle_pins import check_survey_stats, survey_plate
    c, k, cam, H, W, spp = survey_plate()
    ctx = ra.Context(0)
    try:
        ctx.upload(ra.HostScene([c], [k]))
        ctx.rng_seed(5489)
        ctx.render(*cam, H, W, spp, counters=True)
        st = ctx.stats()
    finally:
        ctx.close()
    check_survey_stats(st["nodes_visited"], st["tris_tested"], st["rays_hit"], st["rays"])
    w = orc.Scene([c], [k]).render(*cam, H, W, spp, rng=orc.Rng())
    assert (st["nodes_visited"], st["tris_tested"], st["rays_hit"]) == (w["nodes"], w["tris"], w["hits"])
