"""Randomised differential test: the HIP path against the oracle on seeded random scenes, cameras and
sampling settings chosen to reach the kernel's rarely taken paths -- coordinates around 1e120 (slab
products overflow: the exact compare-select slab test instead of the FAST one), cameras sitting exactly on
box planes with axis-aligned views (0*inf = NaN in the slab test), mixed-sign wave octants, sheets of
coincident triangles (ties), degenerate (zero-area) triangles, many small meshes (TLAS quirk), odd image
sizes and sample counts.  RT_FUZZ_CASES scales the number of cases (default 24, a few seconds)."""
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
def ctx():
    c = ra.Context(0)
    yield c
    c.close()


def random_case(rng):
    kind = rng.integers(0, 6)
    scale = 1.0
    coords, colors = [], []
    if kind == 0:  # soup at a huge or tiny scale
        scale = float(10.0 ** rng.choice([-3, 0, 3, 120]))
        c, k = meshgen.random_soup(int(rng.integers(50, 3000)), seed=int(rng.integers(1 << 30)), extent=4.0, size=0.7)
        coords, colors = [c * scale], [k]
    elif kind == 1:  # integer lattice of axis-aligned quads, camera on lattice planes
        n = int(rng.integers(2, 7))
        tris = []
        for ix in range(-n, n):
            for iy in range(-n, n):
                z = -float(rng.integers(1, 4))
                tris.append([[ix, iy, z], [ix + 1, iy, z], [ix, iy + 1, z]])
                tris.append([[ix + 1, iy, z], [ix + 1, iy + 1, z], [ix, iy + 1, z]])
        c = np.array(tris, float)[None]
        coords, colors = [c], [rng.uniform(0.1, 0.9, (1, c.shape[1], 3))]
    elif kind == 2:  # many small meshes
        for m in range(int(rng.integers(2, 12))):
            c, k = meshgen.uv_sphere(int(rng.integers(3, 9)), int(rng.integers(4, 12)), radius=float(rng.uniform(0.3, 1.2)),
                                     center=tuple(rng.uniform(-3, 3, 2)) + (float(rng.uniform(-8, -3)),))
            coords.append(c)
            colors.append(k * float(rng.uniform(0.1, 1.0)))
    elif kind == 3:  # coincident sheets + degenerate triangles
        c, k = meshgen.plate(int(rng.integers(3, 30)), amp=0.0)
        c2 = c.copy()
        c2[0, ::3, 2, :] = c2[0, ::3, 1, :]  # zero-area triangles
        coords, colors = [c, c.copy(), c2], [k, k * 0.5, k * 0.25]
    elif kind == 4:  # deformed plate, grazing camera
        c, k = meshgen.plate(int(rng.integers(8, 80)), amp=float(rng.uniform(0.0, 0.3)))
        coords, colors = [c], [k]
    else:  # randomly oriented cloud all around the camera
        n = int(rng.integers(100, 4000))
        cc = rng.uniform(-5, 5, (n, 3))
        cc = cc[np.linalg.norm(cc, axis=1) > 0.8]
        arr = cc[:, None, :] + rng.uniform(-0.6, 0.6, (cc.shape[0], 3, 3))
        coords, colors = [np.ascontiguousarray(arr[None])], [rng.uniform(0.0, 1.0, (1, cc.shape[0], 3))]
    W, H = int(rng.integers(1, 70)), int(rng.integers(1, 50))
    spp = int(rng.choice([1, 2, 3, 5, 16, 33, 64, 100, 128]))
    if kind == 1:
        center = tuple(float(x) for x in rng.integers(-2, 3, 3))
        target = tuple(np.array(center) + np.array([0.0, 0.0, -1.0]))
        fov = 90.0
    elif kind == 5:
        center = (0.0, 0.0, 0.0)
        target = tuple(rng.normal(size=3))
        fov = float(rng.uniform(30, 120))
    else:
        center = tuple(rng.uniform(-0.5, 0.5, 3))
        target = tuple(np.array(center) + np.array([rng.uniform(-0.4, 0.4), rng.uniform(-0.4, 0.4), -1.0]))
        fov = float(rng.uniform(30, 110))
    cam = meshgen.camera_arrays(W, H, center=center, target=target, fov=fov)
    if scale != 1.0:  # the camera set-up itself would overflow at 1e120: scale its three arrays instead
        cam = tuple(np.ascontiguousarray(np.asarray(a, np.float64) * scale) for a in cam)
    pos = int(rng.integers(0, 50000)) * 4
    return coords, colors, cam, H, W, spp, pos, kind


@pytest.mark.parametrize("chunk", range(4))
def test_random_scenes_match_the_oracle(ctx, chunk):
    n_cases = int(os.environ.get("RT_FUZZ_CASES", "24"))
    per = (n_cases + 3) // 4
    rng = np.random.default_rng(9000 + chunk)
    kinds_hit = 0
    for case in range(per):
        coords, colors, cam, H, W, spp, pos, kind = random_case(rng)
        host = ra.HostScene(coords, colors, 0)
        ctx.upload(host)
        g = orc.Rng(5489)
        g.discard(pos)
        want = orc.Scene(coords, colors, 0).render(*cam, H, W, spp, rng=g)
        ctx.rng_seed(5489)
        ctx.rng_seek(pos)
        ctx.render(*cam, H, W, spp, hit_ids=True, hit_t=True)
        got = ctx.download()
        mesh, elem = host.map_prims(got["hit_prim"])
        tag = (chunk, case, kind, H, W, spp)
        assert np.array_equal(mesh, want["hit_mesh"]), tag
        assert np.array_equal(elem, want["hit_elem"]), tag
        assert np.array_equal(bits(got["hit_t"]), bits(want["hit_t"])), tag
        assert np.array_equal(bits(got["intensity"]), bits(want["intensity"])), tag
        assert np.array_equal(got["rgb"], want["rgb"]), tag
        kinds_hit += int((want["hit_elem"] >= 0).any())
    assert kinds_hit >= per // 3
