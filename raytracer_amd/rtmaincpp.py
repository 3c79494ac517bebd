"""Drop-in for the reference's nanobind module ``rtmaincpp``.

``cpp_render_scene`` keeps the ten positional arguments of the reference's export
(cpp/rtmain.cpp:29-38, exported at :157-160, called from rtmain.py:29) and forwards them to the
C ABI ``rt_render_scene`` of librtgpu.so, which runs the whole job on the GPU and writes
``<out>/rtimage_<t>_cam<c>.ppm``.  Two ways to put it under the reference's unmodified
``from rtmaincpp import cpp_render_scene`` (rtmain.py:4), both covered by tests/test_dropin.py:
  * ``sys.path.insert(0, "<repo>/dropin")`` -- that directory holds ONLY an ``rtmaincpp`` module, so the
    reference's own rtmain / rtscene / rtcamera keep resolving to the reference's files;
  * ``sys.path.insert(0, "<repo>/raytracer_amd")`` -- ``rtmaincpp`` AND the mirrored rtmain / rtscene /
    rtcamera / rtsimdataloader(_new) resolve here (for machines without pyvale); each of them then
    aliases itself to its ``raytracer_amd.*`` twin, so there is one set of classes either way.

Frame sharding over several processes (one per GPU) is opt-in: with RT_SHARD_FRAMES=1 rank RANK of
WORLD_SIZE renders whole timesteps t % WORLD_SIZE == RANK (all cameras of a timestep, one host BVH build)
on GPU LOCAL_RANK; with fewer timesteps than ranks, single frames f = t*n_cameras + c with
f % WORLD_SIZE == RANK (the schedule is rt_shard_frames).
"""
if not __package__:  # imported as a TOP-LEVEL module (this directory on sys.path, the reference's own layout):
    import importlib as _il  # become an alias of the package module so that there is one set of classes
    import os as _os
    import sys as _sys
    _root = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
    if _root not in _sys.path:
        _sys.path.append(_root)
    _sys.modules[__name__] = _il.import_module("raytracer_amd." + __name__)
import ctypes as C
import os

import numpy as np

from raytracer_amd import _lib as L  # noqa: E402
from raytracer_amd._lib import lib  # noqa: E402


def cpp_render_scene(image_height, image_width, number_of_samples, output_directory, timestep_count,
                     scene_coords_expanded, scene_face_colors, camera_centers, pixel_00_centers,
                     matrix_pixel_spacings):
    coords = [np.ascontiguousarray(c, np.float64) for c in scene_coords_expanded]
    colors = [np.ascontiguousarray(c, np.float64) for c in scene_face_colors]
    if len(coords) != len(colors):
        raise TypeError("cpp_render_scene: coordinate and colour lists differ in length")
    for m, (c, k) in enumerate(zip(coords, colors)):
        if c.ndim != 4 or c.shape[2:] != (3, 3) or k.ndim != 3 or k.shape[2] != 3 or c.shape[:2] != k.shape[:2]:
            raise TypeError(f"cpp_render_scene: mesh {m}: need coords (T,E,3,3) and colours (T,E,3), "
                            f"got {c.shape} and {k.shape}")
        if c.shape[0] < timestep_count:
            raise ValueError(f"cpp_render_scene: mesh {m} has {c.shape[0]} timesteps, {timestep_count} requested")
    n_cam = len(camera_centers)
    if len(pixel_00_centers) != n_cam or len(matrix_pixel_spacings) != n_cam:
        raise TypeError("cpp_render_scene: camera lists differ in length")
    cams = np.ascontiguousarray(np.array(camera_centers, np.float64).reshape(n_cam, 3))
    p00s = np.ascontiguousarray(np.array(pixel_00_centers, np.float64).reshape(n_cam, 3))
    sps = np.ascontiguousarray(np.array(matrix_pixel_spacings, np.float64).reshape(n_cam, 6))
    n = len(coords)
    cp = (C.c_void_p * max(n, 1))(*[c.ctypes.data for c in coords])
    kp = (C.c_void_p * max(n, 1))(*[k.ctypes.data for k in colors])
    counts = np.array([k.shape[1] for k in colors], np.int64)
    rank, world, device = 0, 1, int(os.environ.get("RT_DEVICE", "0"))
    if os.environ.get("RT_SHARD_FRAMES") == "1":
        rank = int(os.environ.get("RANK", "0"))
        world = int(os.environ.get("WORLD_SIZE", "1"))
        device = int(os.environ.get("LOCAL_RANK", str(device)))
    rc = lib.rt_render_scene(int(image_height), int(image_width), int(number_of_samples),
                             os.fspath(output_directory).encode(), int(timestep_count), n, cp, kp, L.ptr(counts),
                             n_cam, L.ptr(cams), L.ptr(p00s), L.ptr(sps), device, rank, world)
    L.check(rc, "rt_render_scene")
    return None
