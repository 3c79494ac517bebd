"""What ONE rt_render_scene call costs in a fresh process (what a user of the reference's render_scene sees): the
reference's own kind of job -- one small mesh, 400x225, 50 spp, 100 timesteps -- timed from the call to its return,
with nothing warmed up (no context, no kernels loaded, no polynomials, no buffers).
Usage: [RT_DRIVER_TIMING=2] python profiles/tools/cold_job.py [T [n]]"""
import ctypes as C
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from raytracer_amd import _lib as L, meshgen  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n = int(sys.argv[2]) if len(sys.argv) > 2 else 32
H, W, spp = 225, 400, 50
c, k = meshgen.plate(n, side=5.0, z0=-4.0, T=T, sway=0.01)
coords, colors = np.ascontiguousarray(c), np.ascontiguousarray(k)
cam = meshgen.camera_arrays(W, H, center=(0.0, 0.0, 0.0), target=(0.0, 0.0, -4.0), fov=80)
cc, p00, sp = (np.ascontiguousarray(np.array(x, np.float64).reshape(1, -1)) for x in cam)
cp, kp = (C.c_void_p * 1)(coords.ctypes.data), (C.c_void_p * 1)(colors.ctypes.data)
counts = np.array([colors.shape[1]], np.int64)
out_dir = tempfile.mkdtemp(prefix="rtcold_")
for label in ("cold (first call of the process)", "warm (second call)"):
    sys.stdout.flush()
    saved, devnull = os.dup(1), os.open(os.devnull, os.O_WRONLY)
    os.dup2(devnull, 1)
    t0 = time.perf_counter()
    try:
        rc = L.lib.rt_render_scene(H, W, spp, out_dir.encode(), T, 1, cp, kp, L.ptr(counts), 1, L.ptr(cc), L.ptr(p00), L.ptr(sp), 0, 0, 1)
    finally:
        el = time.perf_counter() - t0
        os.dup2(saved, 1)
        os.close(saved)
        os.close(devnull)
    L.check(rc, "rt_render_scene")
    print(f"{label}: {T} frames of {W}x{H}x{spp}, {2 * n * n} triangles: {el * 1e3:.1f} ms ({T / el:.0f} frames/s)", flush=True)
for f in os.listdir(out_dir):
    os.remove(os.path.join(out_dir, f))
os.rmdir(out_dir)
