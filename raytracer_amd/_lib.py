"""ctypes binding of librtgpu.so (C ABI: include/rtgpu.h).

The shared library is built in-tree by ``__graft_entry__.build()`` / ``make -C raytracer_amd/csrc``.
There is no fallback: if the library is missing, importing this module raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librtgpu.so")

RT_OK = 0
RT_SAMPLER_MT19937 = 0
RT_SAMPLER_EXPLICIT = 1
RT_SAMPLER_PHILOX = 2
RT_FLAG_HIT_IDS = 1
RT_FLAG_COUNTERS = 2
RT_FLAG_HIT_T = 4
RT_FLAG_FP32 = 8
RT_STATIC_SCRATCH_BYTES = 64 << 20
RT_E_BUSY = -7


class RtError(RuntimeError):
    pass


class SceneInfo(C.Structure):
    _fields_ = [("n_meshes", C.c_int32), ("n_tlas_nodes", C.c_int32), ("n_blas_nodes", C.c_int64),
                ("n_tris", C.c_int64), ("n_dev_nodes", C.c_int64), ("stack_need", C.c_int32),
                ("max_leaf_tris", C.c_int32), ("midpoint_fallbacks", C.c_int64), ("build_seconds", C.c_double)]


class RenderParams(C.Structure):
    _fields_ = [("camera_center", C.c_double * 3), ("pixel_00_center", C.c_double * 3),
                ("pixel_spacing", C.c_double * 6), ("image_height", C.c_int32), ("image_width", C.c_int32),
                ("samples", C.c_int32), ("row_begin", C.c_int32), ("row_end", C.c_int32), ("sampler", C.c_int32),
                ("flags", C.c_uint32), ("jitter", C.c_void_p)]


class Stats(C.Structure):
    _fields_ = [("render_kernel_ms", C.c_double), ("jitter_kernel_ms", C.c_double), ("host_jump_ms", C.c_double),
                ("rays", C.c_uint64), ("nodes_visited", C.c_uint64), ("tris_tested", C.c_uint64),
                ("rays_hit", C.c_uint64), ("wave_node_fetches", C.c_uint64), ("wave_tri_fetches", C.c_uint64),
                ("render_kernel_ms_total", C.c_double), ("render_launches_timed", C.c_uint64),
                ("jitter_streams", C.c_uint64), ("finest_cut_level", C.c_int64)]


# name -> (restype, argtypes).  Every symbol declared in include/rtgpu.h is listed here;
# tests/test_abi.py checks the two against each other.
_VP = C.c_void_p
_PROTOS = {
    "rt_last_error": (C.c_char_p, []),
    "rt_version": (C.c_char_p, []),
    "rt_device_count": (C.c_int, []),
    "rt_host_threads": (C.c_int, []),
    "rt_reload_options": (C.c_int, []),
    "rt_option_names": (C.c_char_p, []),
    "rt_partition_selftest": (C.c_int64, [C.c_int64, C.c_uint64, C.c_int]),
    "rt_mt_phi": (C.c_int, [C.c_void_p, C.c_int]),
    "rt_host_alloc": (_VP, [C.c_size_t]),
    "rt_host_free": (None, [_VP]),
    "rt_scene_build": (_VP, [C.c_int, C.POINTER(_VP), C.POINTER(_VP), _VP, C.c_int, C.c_int]),
    "rt_scene_free": (None, [_VP]),
    "rt_scene_info": (C.c_int, [_VP, C.POINTER(SceneInfo)]),
    "rt_scene_tlas": (C.c_int, [_VP, _VP, _VP, _VP, _VP]),
    "rt_scene_slot_info": (C.c_int, [_VP, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                     C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "rt_scene_slot_dev_root": (C.c_int64, [_VP, C.c_int]),
    "rt_scene_slot_nodes": (C.c_int, [_VP, C.c_int, _VP, _VP, _VP, _VP]),
    "rt_scene_slot_indices": (C.c_int, [_VP, C.c_int, _VP]),
    "rt_scene_prim_tables": (C.c_int, [_VP, _VP, _VP]),
    "rt_create": (_VP, [C.c_int]),
    "rt_destroy": (None, [_VP]),
    "rt_upload_scene": (C.c_int, [_VP, _VP]),
    "rt_rng_seed": (C.c_int, [_VP, C.c_uint32]),
    "rt_rng_seek": (C.c_int, [_VP, C.c_uint64]),
    "rt_rng_tell": (C.c_uint64, [_VP]),
    "rt_rng_jitter": (C.c_int, [_VP, C.c_int64, C.c_int, _VP]),
    "rt_rng_state": (C.c_int, [_VP, _VP]),
    "rt_mt19937_state_at": (C.c_int, [C.c_uint32, C.c_uint64, _VP]),
    "rt_mt19937_stream_states": (C.c_int, [C.c_uint32, C.c_uint64, C.c_uint64, C.c_int, _VP]),
    "rt_render": (C.c_int, [_VP, C.POINTER(RenderParams)]),
    "rt_render_static": (C.c_int, [_VP, C.POINTER(RenderParams), _VP, _VP, _VP, _VP]),
    "rt_download": (C.c_int, [_VP, _VP, _VP, _VP, _VP]),
    "rt_last_stats": (C.c_int, [_VP, C.POINTER(Stats)]),
    "rt_sync": (C.c_int, [_VP]),
    "rt_last_cuts": (C.c_int, [_VP, C.c_int, _VP, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                               C.POINTER(C.c_int32)]),
    "rt_tile_cuts_host": (C.c_int, [_VP, C.POINTER(RenderParams), C.c_int, _VP, C.c_int64, C.POINTER(C.c_int32),
                                    C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "rt_device_outputs": (C.c_int, [_VP, C.POINTER(_VP), C.POINTER(_VP), C.POINTER(_VP)]),
    "rt_trace_rays": (C.c_int, [_VP, C.c_int64, _VP, _VP, _VP, _VP, _VP, _VP]),
    "rt_render_scene": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_int, C.POINTER(_VP),
                                  C.POINTER(_VP), _VP, C.c_int, _VP, _VP, _VP, C.c_int, C.c_int, C.c_int]),
    "rt_shard_frames": (C.c_int64, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_int64, _VP, _VP]),
    "rt_driver_stream_position": (C.c_uint64, [C.c_int]),
    "rt_shard_rows": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.POINTER(C.c_int),
                                C.POINTER(C.c_int), C.POINTER(C.c_uint64)]),
    "rt_write_ppm": (C.c_int, [C.c_char_p, C.c_int, C.c_int, _VP]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). raytracer_amd has no CPU or eager fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


class _Lib:
    """The loaded library.  librtgpu.so takes its RT_* options from the environment once (rt_create) and again only on
    rt_reload_options(); scripts and tests written against this package switch such variables between two calls of one
    process (os.environ[...] = ...), so every call through this object first re-reads them when any RT_* variable has
    changed since the last call: ~30 dictionary look-ups (about 2 us) per call."""

    def __init__(self, raw):
        self._raw = raw
        self._seen = None
        names = raw.rt_option_names().decode().split(",")
        # os.environ keeps its data in a dict keyed by encoded names (CPython); fall back to the public mapping
        data = getattr(os.environ, "_data", None)
        if isinstance(data, dict) and all(isinstance(k, bytes) for k in list(data)[:1]):
            self._env, self._keys = data, [os.fsencode(n) for n in names]
        else:
            self._env, self._keys = os.environ, names

    def _sync(self):
        env = self._env
        now = [env.get(k) for k in self._keys]
        if now != self._seen:
            self._raw.rt_reload_options()
            self._seen = now

    def __getattr__(self, name):
        fn = getattr(self._raw, name)
        if name in ("rt_last_error", "rt_reload_options"):
            call = fn
        else:
            def call(*args, _fn=fn):
                self._sync()
                return _fn(*args)
        self.__dict__[name] = call
        return call


lib = _Lib(_load())


def check(rc, what=""):
    if rc != RT_OK:
        raise RtError(f"{what} failed ({rc}): {lib.rt_last_error().decode()}")


def ptr(a):
    """void* of a numpy array (or None)."""
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a
