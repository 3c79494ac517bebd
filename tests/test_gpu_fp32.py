"""fp32 FAST PATH (RT_FLAG_FP32) -- not the parity path.  BASELINE.json north_star: "an fp32 fast path
reported separately with its tolerance".  These tests state that tolerance against the f64 oracle on the
same samples: the fraction of samples whose hit triangle differs (rays grazing an edge) and the pixel
intensity error; the exact f64 path is covered by test_gpu_parity.py."""
import numpy as np
import pytest

import orc
import raytracer_amd as ra
from raytracer_amd import meshgen

pytestmark = pytest.mark.gpu

FP32_HIT_MISMATCH_MAX = 2e-3   # fraction of samples with a different hit triangle
FP32_INTENSITY_TOL = 2e-2      # max abs pixel intensity difference (a mismatched sample moves a pixel by <= 1/spp)
FP32_INTENSITY_MEAN_TOL = 2e-4  # mean abs pixel intensity difference


@pytest.fixture(scope="module")
def ctx():
    c = ra.Context(0)
    yield c
    c.close()


def compare(ctx, coords, colors, cam, H, W, spp):
    host = ra.HostScene(coords, colors, 0)
    ctx.upload(host)
    want = orc.Scene(coords, colors, 0).render(*cam, H, W, spp, rng=orc.Rng(5489))
    ctx.rng_seed(5489)
    ctx.render(*cam, H, W, spp, hit_ids=True, hit_t=True, fp32=True)
    got = ctx.download()
    assert ctx.rng_tell() == H * W * spp * 4
    mesh, elem = host.map_prims(got["hit_prim"])
    mism = float(((mesh != want["hit_mesh"]) | (elem != want["hit_elem"])).mean())
    d = np.abs(got["intensity"] - want["intensity"])
    same = (mesh == want["hit_mesh"]) & (elem == want["hit_elem"]) & (elem >= 0)
    rel_t = np.abs(got["hit_t"][same] - want["hit_t"][same]) / want["hit_t"][same]
    return mism, float(d.max()), float(d.mean()), float(rel_t.max()) if rel_t.size else 0.0, want


@pytest.mark.parametrize("spp", [1, 16, 64])
def test_fp32_plate_within_stated_tolerance(ctx, spp):
    c, k = meshgen.plate(96)
    cam = meshgen.camera_arrays(64, 48)
    mism, dmax, dmean, rt_, want = compare(ctx, [c], [k], cam, 48, 64, spp)
    assert (want["hit_elem"] >= 0).mean() > 0.5
    assert mism <= FP32_HIT_MISMATCH_MAX, mism
    assert dmax <= max(FP32_INTENSITY_TOL, 1.0 / spp * 0.05) or spp == 1
    assert dmean <= FP32_INTENSITY_MEAN_TOL * (64 / spp if spp < 64 else 1)
    assert rt_ < 1e-4


def test_fp32_multi_mesh_and_octants(ctx):
    rng = np.random.default_rng(5)
    coords, colors = [], []
    for m in range(5):
        c, k = meshgen.uv_sphere(12, 16, radius=0.9, center=tuple(rng.uniform(-3, 3, 3) + np.array([0, 0, -1.0])))
        coords.append(c)
        colors.append(k * (0.2 + 0.15 * m))
    tot = 0.0
    for t in [(0.3, 0.2, -1.0), (1.0, 0.4, 0.3), (-1.0, -0.5, 0.4), (0.2, 1.0, 0.7), (0.1, -1.0, -0.6), (0.3, 0.2, 1.0)]:
        cam = meshgen.camera_arrays(48, 32, center=(0.0, 0.0, -1.0), target=t, fov=100)
        mism, dmax, dmean, rt_, want = compare(ctx, coords, colors, cam, 32, 48, 8)
        assert mism <= FP32_HIT_MISMATCH_MAX, (t, mism)
        assert dmean <= 1e-3
        tot += (want["hit_elem"] >= 0).mean()
    assert tot > 0.3


def test_fp32_flag_does_not_disturb_the_exact_path(ctx):
    c, k = meshgen.plate(32)
    cam = meshgen.camera_arrays(32, 24)
    host = ra.HostScene([c], [k], 0)
    ctx.upload(host)
    ctx.rng_seed(5489)
    ctx.render(*cam, 24, 32, 4, hit_ids=True)
    a = ctx.download()
    ctx.rng_seed(5489)
    ctx.render(*cam, 24, 32, 4, hit_ids=True, fp32=True)
    ctx.rng_seed(5489)
    ctx.render(*cam, 24, 32, 4, hit_ids=True)
    b = ctx.download()
    assert np.array_equal(a["hit_prim"], b["hit_prim"])
    assert np.array_equal(a["intensity"].view(np.uint64), b["intensity"].view(np.uint64))
