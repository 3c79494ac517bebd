"""Soak test of rt_render_scene at scale: the same job (8 meshes rebuilt per timestep, 2 cameras, 64 spp, large
frames) through the pipelined driver (cuts, prep + cut kernels beside the previous frame) and with RT_NO_PIPELINE=1 /
RT_BUILD_AHEAD=1 / RT_NO_CUT=1 / RT_NO_PREP_OVERLAP=1 in a child process; every P6 file must be byte-identical.  Usage: python profiles/tools/soak_driver.py [T] [size]"""
import ctypes as C
import hashlib
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))


def run(T, size, out_dir):
    from raytracer_amd import _lib as L, meshgen
    H = W = size
    spp = 64
    coords, colors = [], []
    for m in range(8):
        c, k = meshgen.plate(96, side=5.0, x0=-10.0 + 5.0 * (m % 4), y0=-5.0 + 5.0 * (m // 4), z0=-6.0, T=T, sway=0.02)
        coords.append(np.ascontiguousarray(c))
        colors.append(np.ascontiguousarray(k))
    cams = [meshgen.camera_arrays(W, H, center=(-0.3 + 0.6 * i, 0.0, 0.0), target=(0.0, 0.0, -6.0), fov=80) for i in range(2)]
    cc = np.ascontiguousarray(np.array([x[0] for x in cams], np.float64))
    p00 = np.ascontiguousarray(np.array([x[1] for x in cams], np.float64))
    sp = np.ascontiguousarray(np.array([x[2] for x in cams], np.float64).reshape(2, 6))
    cp = (C.c_void_p * 8)(*[c.ctypes.data for c in coords])
    kp = (C.c_void_p * 8)(*[k.ctypes.data for k in colors])
    counts = np.array([k.shape[1] for k in colors], np.int64)
    sys.stdout.flush()
    saved = os.dup(1)
    devnull = os.open(os.devnull, os.O_WRONLY)
    os.dup2(devnull, 1)
    t0 = time.perf_counter()
    try:
        rc = L.lib.rt_render_scene(H, W, spp, out_dir.encode(), T, 8, cp, kp, L.ptr(counts), 2, L.ptr(cc), L.ptr(p00),
                                   L.ptr(sp), 0, 0, 1)
    finally:
        os.dup2(saved, 1)
        os.close(saved)
        os.close(devnull)
    L.check(rc, "rt_render_scene")
    return time.perf_counter() - t0


def digest(d):
    return {f: hashlib.sha256(open(os.path.join(d, f), "rb").read()).hexdigest() for f in sorted(os.listdir(d))}


if __name__ == "__main__":
    if len(sys.argv) > 3 and sys.argv[3] == "child":
        run(int(sys.argv[1]), int(sys.argv[2]), sys.argv[4])
        sys.exit(0)
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    a = tempfile.mkdtemp(prefix="rtsoak_a_")
    b = tempfile.mkdtemp(prefix="rtsoak_b_")
    el = run(T, size, a)
    print(f"pipelined: {2 * T} frames of {size}x{size}x64 in {el:.2f} s ({2 * T / el:.1f} frames/s)", flush=True)
    # the child: no pipelining, one build at a time, the walk from the root (no cuts), prep on the traversal stream
    env = dict(os.environ, RT_NO_PIPELINE="1", RT_BUILD_AHEAD="1", RT_NO_CUT="1", RT_NO_PREP_OVERLAP="1")
    subprocess.run([sys.executable, __file__, str(T), str(size), "child", b], check=True, env=env)
    da, db = digest(a), digest(b)
    assert len(da) == 2 * T, len(da)
    bad = [f for f in da if da[f] != db.get(f)]
    for d in (a, b):
        for f in os.listdir(d):
            os.remove(os.path.join(d, f))
        os.rmdir(d)
    if bad:
        print("DIFFERENT:", bad)
        sys.exit(1)
    print(f"all {len(da)} files identical to the unpipelined root-walk run", flush=True)
