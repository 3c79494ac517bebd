"""rt_render_static -- the entry point that allocates nothing (include/rtgpu.h) -- against rt_render.

Through the C ABI, on the GPU: the same frame rendered by the existing entry point (rt_render with
RT_SAMPLER_EXPLICIT + rt_download) and by rt_render_static (jitter and outputs in caller-owned device memory,
allocated by the test straight from the HIP runtime the library itself is linked to; scratch = the library's
file-scope device array) must agree bit for bit -- intensity, P6
payload, hit ids and the visit counters that live in the scratch array -- on two calls in a row, the second
on a smaller scene and a smaller image (the array keeps the first call's contents).  Also: device memory in
use does not change across the calls, and sizes beyond the array are refused with RT_E_LIMIT.
"""
import ctypes as C

import numpy as np
import pytest

import raytracer_amd as ra
from raytracer_amd import _lib as L
from raytracer_amd import meshgen

pytestmark = pytest.mark.gpu


class Hip:
    """The caller's side of the contract: device memory obtained from the HIP runtime librtgpu.so is linked to
    (same soname, so the already loaded library), not from the library under test."""

    def __init__(self):
        self.rt = C.CDLL("libamdhip64.so.7")
        self.rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.rt.hipFree.argtypes = [C.c_void_p]
        self.rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.rt.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        self.live = []

    def ok(self, rc, what):
        assert rc == 0, f"{what} -> hipError {rc}"

    def upload(self, a):
        a = np.ascontiguousarray(a)
        p = C.c_void_p()
        self.ok(self.rt.hipMalloc(C.byref(p), max(a.nbytes, 1)), "hipMalloc")
        self.live.append(p)
        self.ok(self.rt.hipMemcpy(p, a.ctypes.data_as(C.c_void_p), a.nbytes, 1), "hipMemcpy H2D")
        return p.value

    def download(self, p, shape, dtype):
        out = np.empty(shape, dtype)
        self.ok(self.rt.hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(p), out.nbytes, 2), "hipMemcpy D2H")
        return out

    def free_bytes(self):
        f, t = C.c_size_t(), C.c_size_t()
        self.ok(self.rt.hipMemGetInfo(C.byref(f), C.byref(t)), "hipMemGetInfo")
        return f.value

    def release(self):
        for p in self.live:
            self.rt.hipFree(p)
        self.live = []


def bits(a):
    return np.ascontiguousarray(a, np.float64).view(np.uint64)


def _case(ctx, hip, n_plate, H, W, spp, seed, soup=0, counters=True):
    """Render one frame both ways; returns (existing entry point's outputs + counters, static ones)."""
    coords, colors = meshgen.plate(n_plate, side=10.0)
    cs, ks = [coords], [colors]
    if soup:
        c2, k2 = meshgen.random_soup(soup, seed=seed + 1, extent=4.0, z_range=(-5.0, -3.0))
        cs.append(c2)
        ks.append(k2)
    host = ra.HostScene(cs, ks, 0)
    ctx.upload(host)
    cam = meshgen.camera_arrays(W, H)
    jit = np.random.default_rng(seed).uniform(-0.5, 0.5, (H * W * spp, 2))
    # existing entry point
    ctx.render(*cam, H, W, spp, jitter=jit, hit_ids=True, counters=counters)
    want = ctx.download()
    st = ctx.stats()
    want["cut_level"] = st["finest_cut_level"]
    want["counters"] = tuple(st[k] for k in ("nodes_visited", "tris_tested", "rays_hit", "wave_node_fetches",
                                             "wave_tri_fetches"))
    # the entry point that allocates nothing: caller-owned device memory, pre-filled with sentinels
    d_jit = hip.upload(jit)
    d_int = hip.upload(np.full((H, W), -7.0))
    d_rgb = hip.upload(np.full((H, W, 3), 77, np.uint8))
    d_hit = hip.upload(np.full(H * W * spp, -5, np.int32))
    free0 = hip.free_bytes()
    ctx.render_static(*cam, H, W, spp, d_jit, d_int, d_rgb, d_hit, counters=counters)
    ctx.sync()
    st = ctx.stats()
    free1 = hip.free_bytes()
    got = dict(intensity=hip.download(d_int, (H, W), np.float64), rgb=hip.download(d_rgb, (H, W, 3), np.uint8),
               hit_prim=hip.download(d_hit, H * W * spp, np.int32),
               counters=tuple(st[k] for k in ("nodes_visited", "tris_tested", "rays_hit", "wave_node_fetches",
                                              "wave_tri_fetches")),
               free_delta=free0 - free1, kernel_ms=st["render_kernel_ms"], rays=st["rays"],
               cut_level=st["finest_cut_level"])
    host.close()
    return want, got


def _same(want, got, counters=True):
    assert np.array_equal(bits(got["intensity"]), bits(want["intensity"]))
    assert np.array_equal(got["rgb"], want["rgb"])
    assert np.array_equal(got["hit_prim"], want["hit_prim"])
    assert (got["hit_prim"] >= 0).any()  # the frame does hit the plate
    if counters:
        assert got["counters"] == want["counters"]
        assert got["counters"][0] > 0 and got["counters"][2] > 0
    assert got["free_delta"] == 0, "device memory in use changed across rt_render_static"
    assert got["kernel_ms"] > 0.0


def test_static_entry_point_equals_rt_render_twice_in_a_row_second_call_smaller():
    ctx = ra.Context(0)
    hip = Hip()
    try:
        # first call: 2 BLAS (plate + soup: back faces, overlapping hits), 64 spp -> the one-pixel-per-batch kernel
        want, got = _case(ctx, hip, 96, 72, 104, 64, seed=11, soup=3000)
        _same(want, got)
        assert got["rays"] == 72 * 104 * 64
        # second call, smaller scene AND smaller image, generic instantiation, ragged tiles: the scratch array
        # still holds the first call's records and counters beyond what this call overwrites
        want2, got2 = _case(ctx, hip, 20, 37, 50, 3, seed=12)
        _same(want2, got2)
        assert got2["counters"] != got["counters"]
        # the context's own outputs (rt_download) are still those of the last rt_render, not of the static call
        again = ctx.download()
        assert np.array_equal(bits(again["intensity"]), bits(want2["intensity"]))
    finally:
        hip.release()
        ctx.close()


def test_static_entry_point_near_its_maximum_scene():
    """352,800 triangles: 45 MB of triangle records + 13 MB of nodes, close to the 64 MiB the array holds."""
    ctx = ra.Context(0)
    hip = Hip()
    try:
        want, got = _case(ctx, hip, 420, 64, 64, 16, seed=13)
        _same(want, got)
    finally:
        hip.release()
        ctx.close()


def test_static_entry_point_with_cuts_in_the_scratch_array_and_without_room_for_them():
    """Without visit counters the launch makes the pixel blocks' cuts (tests/conftest.py: for every frame size) and
    runs the counter-free kernel instantiations: the cut tables are carved out of the scratch array BEHIND the
    triangle records -- a wrong offset or size would overrun the records or the end of the array.  Small scene: the
    cuts fit and are used (finest_cut_level >= 0), 64 spp (one pixel per batch) and 5 spp (generic), twice in a row;
    a scene near the maximum with a 1024 x 1024 image: they do not fit and the launch walks from the root (-1).
    Bit-equal to rt_render either way."""
    ctx = ra.Context(0)
    hip = Hip()
    try:
        want, got = _case(ctx, hip, 96, 72, 104, 64, seed=21, soup=3000, counters=False)
        _same(want, got, counters=False)
        assert got["cut_level"] >= 0 and want["cut_level"] >= 0
        hip.release()
        want, got = _case(ctx, hip, 40, 45, 61, 5, seed=22, counters=False)
        _same(want, got, counters=False)
        assert got["cut_level"] >= 0
        hip.release()
        want, got = _case(ctx, hip, 420, 1024, 1024, 4, seed=23, counters=False)
        _same(want, got, counters=False)
        assert got["cut_level"] == -1 and want["cut_level"] >= 0  # rt_render has room for its cuts, the array has not
    finally:
        hip.release()
        ctx.close()


def test_static_entry_point_with_the_one_level_cuts_of_mid_sized_frames(monkeypatch):
    """Frames of 1 M .. 12 M rays get one cut level made by cut_single_kernel (several lanes per block, static LDS
    only): through the allocation-free entry point too, with the level carved out of the scratch array.  Here the
    thresholds are the library's own (tests/conftest.py forces the whole hierarchy otherwise): 150 x 240 x 50 spp =
    1.8 M rays, then a smaller image in the same mode."""
    monkeypatch.delenv("RT_CUT_MIN_RAYS")
    ctx = ra.Context(0)
    hip = Hip()
    try:
        want, got = _case(ctx, hip, 40, 150, 240, 50, seed=31, counters=False)
        _same(want, got, counters=False)
        assert got["cut_level"] >= 0 and want["cut_level"] == got["cut_level"]
        hip.release()
        monkeypatch.setenv("RT_CUT1_MIN_RAYS", "0")
        want, got = _case(ctx, hip, 24, 60, 90, 64, seed=32, counters=False)
        _same(want, got, counters=False)
        assert got["cut_level"] >= 0
    finally:
        hip.release()
        ctx.close()


def test_static_entry_point_refuses_a_second_context_while_the_first_ones_launch_runs():
    """All contexts on a GPU share that GPU's scratch array: a call from another context while a launch into the
    array has not finished returns RT_E_BUSY and launches nothing; after rt_sync of the first context it goes through."""
    a, b = ra.Context(0), ra.Context(0)
    hip = Hip()
    try:
        coords, colors = meshgen.plate(300, side=10.0)  # 180,000 triangles
        host = ra.HostScene([coords], [colors], 0)
        a.upload(host)
        b.upload(host)
        H = W = 2048
        spp = 16  # 67 M rays: tens of milliseconds of GPU work behind the call that returns in microseconds
        cam = meshgen.camera_arrays(W, H)
        d_jit = hip.upload(np.random.default_rng(5).uniform(-0.5, 0.5, (H * W * spp, 2)))
        d_int, d_rgb = hip.upload(np.zeros((H, W))), hip.upload(np.zeros((H, W, 3), np.uint8))
        d_int2, d_rgb2 = hip.upload(np.full((H, W), -7.0)), hip.upload(np.zeros((H, W, 3), np.uint8))
        a.render_static(*cam, H, W, spp, d_jit, d_int, d_rgb)
        with pytest.raises(ra.RtError, match="another context"):
            b.render_static(*cam, H, W, spp, d_jit, d_int2, d_rgb2)
        assert (hip.download(d_int2, (H, W), np.float64) == -7.0).all()  # the refused call wrote nothing
        a.render_static(*cam, H, W, spp, d_jit, d_int, d_rgb)  # the SAME context may queue behind itself
        a.sync()
        first = hip.download(d_int, (H, W), np.float64)
        b.render_static(*cam, H, W, spp, d_jit, d_int2, d_rgb2)
        b.sync()
        assert np.array_equal(bits(hip.download(d_int2, (H, W), np.float64)), bits(first))
        host.close()
    finally:
        hip.release()
        a.close()
        b.close()


def test_static_entry_point_rejects_what_it_cannot_hold_or_do():
    ctx = ra.Context(0)
    try:
        H, W, spp = 16, 16, 2
        cam = meshgen.camera_arrays(W, H)
        p = L.RenderParams()
        p.camera_center[:] = list(cam[0])
        p.pixel_00_center[:] = list(cam[1])
        p.pixel_spacing[:] = list(np.asarray(cam[2]).reshape(6))
        p.image_height, p.image_width, p.samples, p.row_begin, p.row_end = H, W, spp, 0, H
        p.sampler = L.RT_SAMPLER_EXPLICIT
        one = 1 << 20  # never dereferenced: every call below must fail before any launch
        assert L.lib.rt_render_static(ctx.h, p, one, one, one, None) == -3  # RT_E_NO_SCENE
        coords, colors = meshgen.plate(480, side=10.0)  # 460,800 triangles: 59 MB of triangle records + 17 MB of nodes
        host = ra.HostScene([coords], [colors], 0)
        need = 256 + 64 * ((host.info.n_tlas_nodes + 3) // 4 * 4) + 64 * host.info.n_dev_nodes + 128 * host.info.n_tris
        assert need > L.RT_STATIC_SCRATCH_BYTES
        ctx.upload(host)
        assert L.lib.rt_render_static(ctx.h, p, one, one, one, None) == -5  # RT_E_LIMIT
        msg = L.lib.rt_last_error().decode()
        assert "scratch" in msg and str(L.RT_STATIC_SCRATCH_BYTES) in msg and "460800" in msg
        host.close()
        c2, k2 = meshgen.plate(8, side=10.0)
        host = ra.HostScene([c2], [k2], 0)
        ctx.upload(host)
        p.sampler = L.RT_SAMPLER_MT19937
        assert L.lib.rt_render_static(ctx.h, p, one, one, one, None) == -1  # RT_E_INVALID: explicit jitter only
        p.sampler = L.RT_SAMPLER_EXPLICIT
        assert L.lib.rt_render_static(ctx.h, p, None, one, one, None) == -1
        p.flags = L.RT_FLAG_HIT_IDS
        assert L.lib.rt_render_static(ctx.h, p, one, one, one, None) == -1  # hit ids without a buffer
        p.flags = L.RT_FLAG_FP32
        assert L.lib.rt_render_static(ctx.h, p, one, one, one, None) == -1
        host.close()
    finally:
        ctx.close()
