"""Drop-in for the reference's nanobind module ``rtmaincpp``.

``cpp_render_scene`` keeps the ten positional arguments of the reference's export
(cpp/rtmain.cpp:29-38, exported at :157-160, called from rtmain.py:29) and forwards them to the
C ABI ``rt_render_scene`` of librtgpu.so, which runs the whole job on the GPU and writes
``<out>/rtimage_<t>_cam<c>.ppm``.  To use it from the unmodified reference scripts put this
package's directory ahead on sys.path (``import rtmaincpp`` then resolves to this file).

Frame sharding over several processes (one per GPU) is opt-in: with RT_SHARD_FRAMES=1 the frames
f = t*n_cameras + c with f % WORLD_SIZE == RANK are rendered on GPU LOCAL_RANK.
"""
import ctypes as C
import os

import numpy as np

from . import _lib as L
from ._lib import lib


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
