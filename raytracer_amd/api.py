"""Thin object layer over the C ABI (include/rtgpu.h): HostScene (built tree) and Context (one GPU)."""
import ctypes as C

import numpy as np

from . import _lib as L
from ._lib import lib


def _mesh_arrays(coords, colors):
    coords = [L.f64(c) for c in coords]
    colors = [L.f64(c) for c in colors]
    if len(coords) != len(colors) or not coords:
        raise ValueError("need the same, non-zero number of coordinate and colour arrays")
    for m, (c, k) in enumerate(zip(coords, colors)):
        # the reference takes E from face_colors.shape(1) (cpp/rtbvh.cpp:615): (T,E,3,3) / (T,E,3)
        if c.ndim != 4 or c.shape[2:] != (3, 3) or k.ndim != 3 or k.shape[2] != 3:
            raise ValueError(f"mesh {m}: expected coords (T,E,3,3) and colours (T,E,3), got {c.shape} / {k.shape}")
        if c.shape[:2] != k.shape[:2]:
            raise ValueError(f"mesh {m}: coords {c.shape} and colours {k.shape} disagree on (T,E)")
    n = len(coords)
    cp = (C.c_void_p * n)(*[c.ctypes.data for c in coords])
    kp = (C.c_void_p * n)(*[k.ctypes.data for k in colors])
    counts = np.array([k.shape[1] for k in colors], np.int64)
    return coords, colors, cp, kp, counts


class _Pinned:
    """Owner of one rt_host_alloc block (page-locked host memory)."""

    def __init__(self, nbytes):
        self.p = lib.rt_host_alloc(nbytes)
        if not self.p:
            raise L.RtError(lib.rt_last_error().decode())

    def __del__(self):
        if getattr(self, "p", None):
            lib.rt_host_free(self.p)
            self.p = None


def pinned_empty(shape, dtype=np.float64):
    """np.empty in page-locked host memory (freed with the array): rt_download into it runs at the link's
    rate instead of the ~10 GB/s of a pageable copy."""
    dtype = np.dtype(dtype)
    n = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
    holder = _Pinned(max(n, 1))
    buf = (C.c_char * max(n, 1)).from_address(holder.p)
    buf._rt_holder = holder  # the numpy array keeps `buf` alive, `buf` keeps the allocation alive
    return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape, dtype=np.int64))).reshape(shape)


class HostScene:
    """Acceleration structure of one timestep, built on the host exactly as the reference's
    build_acceleration_structures (cpp/rtbvh.cpp:590-685) and flattened for the GPU.
    n_threads: 0 = the library's pool (big trees split into tasks: least latency of one build), > 0 = that many
    threads over the meshes, < 0 = the pool with one thread per tree (least CPU time; for callers that keep several
    builds in flight on threads of their own, as rt_render_scene does).  The trees do not depend on it."""

    def __init__(self, coords, colors, timestep=0, n_threads=0):
        self._keep = _mesh_arrays(coords, colors)
        coords, colors, cp, kp, counts = self._keep
        T = coords[0].shape[0]
        if not 0 <= timestep < T:
            raise ValueError(f"timestep {timestep} outside [0,{T})")
        self.h = lib.rt_scene_build(len(coords), cp, kp, L.ptr(counts), int(timestep), int(n_threads))
        if not self.h:
            raise L.RtError(lib.rt_last_error().decode())
        info = L.SceneInfo()
        L.check(lib.rt_scene_info(self.h, C.byref(info)), "rt_scene_info")
        self.info = info
        self._prim = None

    def close(self):
        if getattr(self, "h", None):
            lib.rt_scene_free(self.h)
            self.h = None

    __del__ = close

    def tlas(self):
        n = self.info.n_tlas_nodes
        boxes = np.empty((n, 6))
        cnt, left, mn = (np.empty(n, np.int32) for _ in range(3))
        L.check(lib.rt_scene_tlas(self.h, L.ptr(boxes), L.ptr(cnt), L.ptr(left), L.ptr(mn)), "rt_scene_tlas")
        return dict(boxes=boxes, blas_count=cnt, left=left, min_blas_idx=mn)

    def slot(self, s):
        mesh, nn, ne, tb = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int64()
        L.check(lib.rt_scene_slot_info(self.h, s, C.byref(mesh), C.byref(nn), C.byref(ne), C.byref(tb)),
                "rt_scene_slot_info")
        boxes = np.empty((nn.value, 6))
        left = np.empty(nn.value, np.int32)
        count = np.empty(nn.value, np.int64)
        min_idx = np.empty(nn.value, np.int32)
        L.check(lib.rt_scene_slot_nodes(self.h, s, L.ptr(boxes), L.ptr(left), L.ptr(count), L.ptr(min_idx)),
                "rt_scene_slot_nodes")
        idx = np.empty(ne.value, np.int32)
        L.check(lib.rt_scene_slot_indices(self.h, s, L.ptr(idx)), "rt_scene_slot_indices")
        return dict(mesh=mesh.value, boxes=boxes, left=left, count=count, min_idx=min_idx, indices=idx,
                    tri_base=tb.value, dev_root=int(lib.rt_scene_slot_dev_root(self.h, s)))

    def prim_tables(self):
        """prim id -> (mesh index, element index)."""
        if self._prim is None:
            n = self.info.n_tris
            pm, pe = np.empty(n, np.int32), np.empty(n, np.int32)
            L.check(lib.rt_scene_prim_tables(self.h, L.ptr(pm), L.ptr(pe)), "rt_scene_prim_tables")
            self._prim = (pm, pe)
        return self._prim

    def map_prims(self, prim):
        """Per-sample prim ids -> (mesh, elem) arrays, -1 for misses."""
        pm, pe = self.prim_tables()
        prim = np.asarray(prim)
        hit = prim >= 0
        mesh = np.full(prim.shape, -1, np.int32)
        elem = np.full(prim.shape, -1, np.int32)
        mesh[hit] = pm[prim[hit]]
        elem[hit] = pe[prim[hit]]
        return mesh, elem


def tile_cuts_host(scene: HostScene, cam, p00, spacing, H, W, spp, level=None, rows=None):
    """The cuts rt_render makes for this scene / camera / image at `level` (blocks of 2^level pixels; default: the
    finest level), computed by the host twin of the cut kernel: (words[n_blocks, 32] uint32, tiles_x, (lo, top))."""
    p = L.RenderParams()
    p.camera_center[:] = list(L.f64(cam, 3))
    p.pixel_00_center[:] = list(L.f64(p00, 3))
    p.pixel_spacing[:] = list(L.f64(spacing, 6))
    p.image_height, p.image_width, p.samples = int(H), int(W), int(spp)
    p.row_begin, p.row_end = (0, int(H)) if rows is None else (int(rows[0]), int(rows[1]))
    tx, lo, top = C.c_int32(), C.c_int32(), C.c_int32()
    lib.rt_tile_cuts_host(scene.h, C.byref(p), -1, None, 0, None, C.byref(lo), C.byref(top))  # query the levels
    level = lo.value if level is None else int(level)
    n = lib.rt_tile_cuts_host(scene.h, C.byref(p), level, None, 0, C.byref(tx), None, None)
    if n < 0:
        raise L.RtError(lib.rt_last_error().decode())
    words = np.zeros((n, 32), np.uint32)
    L.check(min(0, lib.rt_tile_cuts_host(scene.h, C.byref(p), level, L.ptr(words), words.size, None, None, None)),
            "rt_tile_cuts_host")
    return words, tx.value, (lo.value, top.value)


class Context:
    """One GPU.  Raises if no HIP device is usable: there is no CPU path."""

    def __init__(self, device=0):
        self.h = lib.rt_create(int(device))
        if not self.h:
            raise L.RtError(lib.rt_last_error().decode())
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            lib.rt_destroy(self.h)
            self.h = None

    __del__ = close

    def upload(self, scene: HostScene):
        L.check(lib.rt_upload_scene(self.h, scene.h), "rt_upload_scene")

    # ---- sampler ----
    def rng_seed(self, seed=5489):
        L.check(lib.rt_rng_seed(self.h, seed), "rt_rng_seed")

    def rng_seek(self, pos):
        L.check(lib.rt_rng_seek(self.h, int(pos)), "rt_rng_seek")

    def rng_tell(self):
        return int(lib.rt_rng_tell(self.h))

    def rng_jitter(self, n, n_streams=4):
        out = np.empty(int(n), np.float64)
        L.check(lib.rt_rng_jitter(self.h, int(n), int(n_streams), L.ptr(out)), "rt_rng_jitter")
        return out

    def rng_state(self):
        out = np.empty(624, np.uint32)
        L.check(lib.rt_rng_state(self.h, L.ptr(out)), "rt_rng_state")
        return out

    # ---- render ----
    def render(self, cam, p00, spacing, H, W, spp, *, rows=None, jitter=None, hit_ids=False, hit_t=False,
               counters=False, fp32=False, philox=False):
        """fp32=True selects the fp32 FAST PATH (RT_FLAG_FP32): not bit-exact, reported separately.
        philox=True selects the counter-based fast sampler (RT_SAMPLER_PHILOX): other samples than the reference."""
        p = L.RenderParams()
        p.camera_center[:] = list(L.f64(cam, 3))
        p.pixel_00_center[:] = list(L.f64(p00, 3))
        p.pixel_spacing[:] = list(L.f64(spacing, 6))
        p.image_height, p.image_width, p.samples = int(H), int(W), int(spp)
        p.row_begin, p.row_end = (0, int(H)) if rows is None else (int(rows[0]), int(rows[1]))
        p.flags = (L.RT_FLAG_HIT_IDS if hit_ids else 0) | (L.RT_FLAG_HIT_T if hit_t else 0) | (
            L.RT_FLAG_COUNTERS if counters else 0) | (L.RT_FLAG_FP32 if fp32 else 0)
        keep = None
        if jitter is not None:
            keep = L.f64(jitter).reshape(-1)
            need = (p.row_end - p.row_begin) * W * spp * 2
            if keep.size != need:
                raise ValueError(f"jitter needs {need} values, got {keep.size}")
            p.sampler = L.RT_SAMPLER_EXPLICIT
            p.jitter = keep.ctypes.data
        else:
            p.sampler = L.RT_SAMPLER_PHILOX if philox else L.RT_SAMPLER_MT19937
        L.check(lib.rt_render(self.h, C.byref(p)), "rt_render")
        self._shape = (p.row_end - p.row_begin, int(W), int(spp))
        self._flags = p.flags

    def render_static(self, cam, p00, spacing, H, W, spp, d_jitter, d_intensity, d_rgb, d_hit_prim=None, *, rows=None,
                      counters=False):
        """rt_render_static: the packet render with NO allocation -- jitter and outputs are raw DEVICE addresses
        (ints) of caller-owned memory, scratch is the library's file-scope device array.  Asynchronous: sync()
        before reading the outputs."""
        p = L.RenderParams()
        p.camera_center[:] = list(L.f64(cam, 3))
        p.pixel_00_center[:] = list(L.f64(p00, 3))
        p.pixel_spacing[:] = list(L.f64(spacing, 6))
        p.image_height, p.image_width, p.samples = int(H), int(W), int(spp)
        p.row_begin, p.row_end = (0, int(H)) if rows is None else (int(rows[0]), int(rows[1]))
        p.flags = (L.RT_FLAG_HIT_IDS if d_hit_prim else 0) | (L.RT_FLAG_COUNTERS if counters else 0)
        p.sampler = L.RT_SAMPLER_EXPLICIT
        L.check(lib.rt_render_static(self.h, C.byref(p), C.c_void_p(d_jitter), C.c_void_p(d_intensity), C.c_void_p(d_rgb),
                                     C.c_void_p(d_hit_prim) if d_hit_prim else None), "rt_render_static")

    def download(self, intensity=True, rgb=True, hit_ids=None, hit_t=None, pinned=True):
        """Results of the last render as numpy arrays (in page-locked memory unless pinned=False)."""
        rows, W, spp = self._shape
        def empty(shape, dtype=np.float64):
            if pinned:
                try:
                    return pinned_empty(shape, dtype)
                except L.RtError:  # no page-locked memory left (per-sample outputs can be several GB): pageable copy
                    pass
            return np.empty(shape, dtype)

        hit_ids = bool(self._flags & L.RT_FLAG_HIT_IDS) if hit_ids is None else hit_ids
        hit_t = bool(self._flags & L.RT_FLAG_HIT_T) if hit_t is None else hit_t
        out = {}
        a = empty((rows, W)) if intensity else None
        b = empty((rows, W, 3), np.uint8) if rgb else None
        c = empty(rows * W * spp, np.int32) if hit_ids else None
        d = empty(rows * W * spp) if hit_t else None
        L.check(lib.rt_download(self.h, L.ptr(a), L.ptr(b), L.ptr(c), L.ptr(d)), "rt_download")
        if a is not None:
            out["intensity"] = a
        if b is not None:
            out["rgb"] = b
        if c is not None:
            out["hit_prim"] = c
        if d is not None:
            out["hit_t"] = d
        return out

    def last_cuts(self, level=None):
        """Cuts of the last render at `level` (default: the finest): (words[n_blocks, 32] uint32, tiles_x, (lo, top));
        n_blocks = 0 if the render made none."""
        tx, lo, top = C.c_int32(), C.c_int32(), C.c_int32()
        n = lib.rt_last_cuts(self.h, -1, None, 0, None, C.byref(lo), C.byref(top))  # query the levels
        if n == 0:
            return np.zeros((0, 32), np.uint32), 0, (0, 0)
        level = lo.value if level is None else int(level)
        n = lib.rt_last_cuts(self.h, level, None, 0, C.byref(tx), None, None)
        if n < 0:
            raise L.RtError(lib.rt_last_error().decode())
        words = np.zeros((n, 32), np.uint32)
        L.check(min(0, lib.rt_last_cuts(self.h, level, L.ptr(words), words.size, None, None, None)), "rt_last_cuts")
        return words, tx.value, (lo.value, top.value)

    def stats(self):
        s = L.Stats()
        L.check(lib.rt_last_stats(self.h, C.byref(s)), "rt_last_stats")
        return {k: getattr(s, k) for k, _ in s._fields_}

    def sync(self):
        L.check(lib.rt_sync(self.h), "rt_sync")

    def trace(self, origins, dirs):
        o = L.f64(origins).reshape(-1, 3)
        d = L.f64(dirs).reshape(-1, 3)
        n = o.shape[0]
        if d.shape[0] != n:
            raise ValueError("origins and dirs differ in length")
        t = np.empty(n)
        bary = np.empty((n, 3))
        col = np.empty((n, 3))
        prim = np.empty(n, np.int32)
        L.check(lib.rt_trace_rays(self.h, n, L.ptr(o), L.ptr(d), L.ptr(t), L.ptr(bary), L.ptr(col), L.ptr(prim)),
                "rt_trace_rays")
        return dict(t=t, bary=bary, color=col, prim=prim)


def mt19937_state_at(pos, seed=5489):
    out = np.empty(624, np.uint32)
    L.check(lib.rt_mt19937_state_at(seed, int(pos), L.ptr(out)), "rt_mt19937_state_at")
    return out


def mt19937_stream_states(base, stride, n, seed=5489):
    out = np.empty((n, 624), np.uint32)
    L.check(lib.rt_mt19937_stream_states(seed, int(base), int(stride), int(n), L.ptr(out)),
            "rt_mt19937_stream_states")
    return out


def shard_frames(timestep_count, n_cameras, rank, world, stream_base=0, frame_words=0):
    """Frames (t*n_cameras + c) rank `rank` of `world` renders, and their RNG stream positions."""
    n = lib.rt_shard_frames(timestep_count, n_cameras, rank, world, stream_base, frame_words, 0, None, None)
    if n < 0:
        raise L.RtError(lib.rt_last_error().decode())
    frames = np.empty(max(n, 1), np.int64)
    pos = np.empty(max(n, 1), np.uint64)
    lib.rt_shard_frames(timestep_count, n_cameras, rank, world, stream_base, frame_words, n, L.ptr(frames), L.ptr(pos))
    return frames[:n], pos[:n]


def shard_rows(H, W, spp, rank, world, frame_base=0):
    """(row_begin, row_end, stream_position) of rank `rank`'s band of ONE frame split over `world` GPUs."""
    r0, r1, pos = C.c_int(), C.c_int(), C.c_uint64()
    L.check(lib.rt_shard_rows(int(H), int(W), int(spp), int(rank), int(world), int(frame_base), C.byref(r0), C.byref(r1),
                              C.byref(pos)), "rt_shard_rows")
    return r0.value, r1.value, pos.value


def render_frame_banded(ctx, cam, p00, spacing, H, W, spp, rank, world, frame_base=0, gather=None, **render_kw):
    """One frame over `world` GPUs: this rank renders its row band (tree replicated, no data-path
    collective) and `gather(band_intensity, band_rgb, row_begin, row_end)` -- e.g. a torch.distributed
    gather to rank 0 -- assembles the framebuffer.  Leaves the context's stream position at the END of
    the frame on every rank, like an unsharded render.  Returns whatever `gather` returns (or the band)."""
    r0, r1, pos = shard_rows(H, W, spp, rank, world, frame_base)
    band = {"intensity": np.empty((0, W)), "rgb": np.empty((0, W, 3), np.uint8)}
    if r1 > r0:
        ctx.rng_seek(pos)
        ctx.render(cam, p00, spacing, H, W, spp, rows=(r0, r1), **render_kw)
        band = ctx.download(hit_ids=False, hit_t=False)
    ctx.rng_seek(int(frame_base) + H * W * spp * 4)
    if gather is None:
        return band, (r0, r1)
    return gather(band["intensity"], band["rgb"], r0, r1)


def write_ppm(path, rgb):
    rgb = np.ascontiguousarray(rgb, np.uint8)
    h, w = rgb.shape[:2]
    L.check(lib.rt_write_ppm(str(path).encode(), w, h, L.ptr(rgb)), "rt_write_ppm")
