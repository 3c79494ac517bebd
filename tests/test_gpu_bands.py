"""Frames whose jitter would not fit the budget are rendered band by band inside rt_render / rt_render_scene
(rt_api.hip render_frame; RT_JITTER_SLOT_BYTES, default 32 GiB per slot): the same frame bit for bit, the same
stream position afterwards, counters and kernel time summed over the bands.  Here the budget is shrunk so that small
frames take that path.
"""
import os

import numpy as np
import pytest

import raytracer_amd as ra
from raytracer_amd import meshgen
from raytracer_amd import _lib as L

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float64).view(np.uint64)


@pytest.fixture(scope="module")
def ctx():
    c = ra.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("spp", [64, 50, 3])
def test_banded_frame_equals_the_frame_in_one_piece(ctx, monkeypatch, spp):
    c0, k0 = meshgen.plate(30, sway=0.1)
    c1, k1 = meshgen.uv_sphere(10, 14, radius=0.8, center=(0.4, 0.3, -2.5))
    host = ra.HostScene([c0, c1], [k0, k1], 0)
    ctx.upload(host)
    H, W = 50, 72
    cam = meshgen.camera_arrays(W, H, center=(0.2, 0.1, 0.3), target=(0.0, 0.0, -2.0), fov=70)
    out = {}
    for mode in ("whole", "bands"):
        if mode == "bands":
            monkeypatch.setenv("RT_JITTER_SLOT_BYTES", str(16 * W * spp * 16))  # 16 rows per band: 4 bands, the last of 2 rows
        else:
            monkeypatch.delenv("RT_JITTER_SLOT_BYTES", raising=False)
        ctx.rng_seed(5489)
        ctx.rng_seek(12345 * 4)
        ctx.render(*cam, H, W, spp, hit_ids=True, hit_t=True, counters=True)
        ctx.sync()
        st = ctx.stats()
        out[mode] = (ctx.download(), st, ctx.rng_tell())
    monkeypatch.delenv("RT_JITTER_SLOT_BYTES", raising=False)
    (a, sa, pa), (b, sb, pb) = out["whole"], out["bands"]
    assert pa == pb == 12345 * 4 + H * W * spp * 4
    assert a["intensity"].shape == b["intensity"].shape == (H, W)
    assert np.array_equal(bits(a["intensity"]), bits(b["intensity"]))
    assert np.array_equal(a["rgb"], b["rgb"])
    assert np.array_equal(a["hit_prim"], b["hit_prim"])
    assert np.array_equal(bits(a["hit_t"]), bits(b["hit_t"]))
    assert (a["hit_prim"] >= 0).any()
    for key in ("rays", "nodes_visited", "tris_tested", "rays_hit"):
        assert sa[key] == sb[key], key
    assert sb["render_kernel_ms"] > 0
    # rows of a frame rendered as the caller's own band (rows=...) are banded again if they must be
    monkeypatch.setenv("RT_JITTER_SLOT_BYTES", str(16 * W * spp * 16))
    ctx.rng_seed(5489)
    ctx.rng_seek(12345 * 4 + 8 * W * spp * 4)
    ctx.render(*cam, H, W, spp, rows=(8, 48), hit_ids=True)
    sub = ctx.download()
    monkeypatch.delenv("RT_JITTER_SLOT_BYTES", raising=False)
    assert np.array_equal(bits(sub["intensity"]), bits(a["intensity"][8:48]))
    assert np.array_equal(sub["hit_prim"].reshape(40, W, spp), a["hit_prim"].reshape(H, W, spp)[8:48])
    host.close()


def test_driver_frames_rendered_in_bands(ctx, monkeypatch, tmp_path):
    """rt_render_scene with frames that need bands (not pipelined then) writes the frames rt_render gives in one
    piece at the same positions of the driver's never-reseeded stream."""
    T = 2
    c, k = meshgen.plate(16, T=T, sway=0.2)
    H, W, spp = 40, 30, 5
    cams = [meshgen.camera_arrays(W, H), meshgen.camera_arrays(W, H, center=(0.5, 0.2, 0.3), target=(0, 0, -2), fov=60)]
    sc = ra.Scene()
    sc.add_mesh(c, None, None, k, T)
    for cam in cams:
        sc.add_camera(*cam)
    base = int(L.lib.rt_driver_stream_position(0))
    monkeypatch.setenv("RT_JITTER_SLOT_BYTES", str(16 * W * spp * 16))  # 16 rows per band
    ra.render_scene(H, W, sc, spp, tmp_path)
    monkeypatch.delenv("RT_JITTER_SLOT_BYTES", raising=False)
    fw = H * W * spp * 4
    assert int(L.lib.rt_driver_stream_position(0)) == base + T * len(cams) * fw
    hdr = f"P6\n{W} {H}\n255\n".encode()
    for t in range(T):
        host = ra.HostScene([c], [k], t)
        ctx.upload(host)
        for ci, cam in enumerate(cams):
            ctx.rng_seed(5489)
            ctx.rng_seek(base + (t * len(cams) + ci) * fw)
            ctx.render(*cam, H, W, spp)
            want = ctx.download()["rgb"]
            assert (tmp_path / f"rtimage_{t}_cam{ci}.ppm").read_bytes() == hdr + want.tobytes(), (t, ci)
        host.close()
