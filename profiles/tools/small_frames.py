"""Frames/s of rt_render_scene in the reference's own regime: small meshes, small images, many timesteps
(BASELINE configs[2], animated_render-style).  Usage: python profiles/tools/small_frames.py [T]"""
import ctypes as C
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from raytracer_amd import _lib as L, meshgen  # noqa: E402


def run_case(name, n, T, H, W, spp, n_cam):
    c, k = meshgen.plate(n, side=5.0, z0=-4.0, T=T, sway=0.01)
    coords, colors = [np.ascontiguousarray(c)], [np.ascontiguousarray(k)]
    cams = [meshgen.camera_arrays(W, H, center=(-0.3 + 0.6 * i, 0.0, 0.0), target=(0.0, 0.0, -4.0), fov=80) for i in range(n_cam)]
    cc = np.ascontiguousarray(np.array([x[0] for x in cams], np.float64))
    p00 = np.ascontiguousarray(np.array([x[1] for x in cams], np.float64))
    sp = np.ascontiguousarray(np.array([x[2] for x in cams], np.float64).reshape(n_cam, 6))
    cp = (C.c_void_p * 1)(coords[0].ctypes.data)
    kp = (C.c_void_p * 1)(colors[0].ctypes.data)
    counts = np.array([colors[0].shape[1]], np.int64)
    out_dir = tempfile.mkdtemp(prefix="rtsmall_")

    def run(n_steps):
        sys.stdout.flush()
        saved = os.dup(1)
        devnull = os.open(os.devnull, os.O_WRONLY)
        os.dup2(devnull, 1)
        try:
            rc = L.lib.rt_render_scene(H, W, spp, out_dir.encode(), n_steps, 1, cp, kp, L.ptr(counts), n_cam, L.ptr(cc),
                                       L.ptr(p00), L.ptr(sp), 0, 0, 1)
        finally:
            os.dup2(saved, 1)
            os.close(saved)
            os.close(devnull)
        L.check(rc, "rt_render_scene")

    run(min(T, 4))
    t0 = time.perf_counter()
    run(T)
    el = time.perf_counter() - t0
    for f in os.listdir(out_dir):
        os.remove(os.path.join(out_dir, f))
    os.rmdir(out_dir)
    print(f"{name}: {2 * n * n} tris, {W}x{H}, {spp} spp, {T} timesteps x {n_cam} cam: {T * n_cam / el:.0f} frames/s "
          f"({el / (T * n_cam) * 1e3:.3f} ms/frame)", flush=True)


if __name__ == "__main__":
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    run_case("tiny", 5, T, 225, 400, 1, 1)
    run_case("small", 32, T, 225, 400, 50, 1)
    run_case("medium", 128, T, 576, 1024, 50, 2)
