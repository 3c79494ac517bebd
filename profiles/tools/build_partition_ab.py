"""Host BVH build of the 5 M-triangle bench scene (4 meshes) and of one 5 M-triangle mesh: parallel closed-form partition
(default) against the serial two-pointer loop (RT_SERIAL_PARTITION=1), per host thread count.  Each setting runs in a
fresh process (the pool width and the knobs are read once).  Usage: python profiles/tools/build_partition_ab.py"""
import os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
CHILD = r'''
import sys, time
sys.path.insert(0, sys.argv[1])
import numpy as np, raytracer_amd as ra
from raytracer_amd import meshgen
which = sys.argv[2]
if which == "c4":
    cs, ks = meshgen.plate_grid(791, grid=(2, 2), side=10.0)
elif which == "one5m":
    c, k = meshgen.plate(1582); cs, ks = [c], [k]
else:
    cs, ks = [], []
    for m in range(8):
        c, k = meshgen.plate(128, side=5.0, x0=-10.0 + 5.0 * (m % 4), y0=-5.0 + 5.0 * (m // 4), z0=-6.0)
        cs.append(np.ascontiguousarray(c)); ks.append(np.ascontiguousarray(k))
ts = []
for _ in range(4):
    t0 = time.time(); h = ra.HostScene(cs, ks, 0); ts.append(time.time() - t0); h.close()
print("%.1f" % (min(ts) * 1e3))
'''
cores = len(os.sched_getaffinity(0))
for which in ("c4", "one5m", "8x32k"):
    for thr in [t for t in (1, 4, 8, 16) if t <= cores]:
        row = []
        for serial in ("", "1"):
            env = dict(os.environ, RT_HOST_THREADS=str(thr))
            if serial:
                env["RT_SERIAL_PARTITION"] = "1"
            row.append(subprocess.check_output([sys.executable, "-c", CHILD, ROOT, which], env=env, text=True).strip())
        print(f"{which:6s} threads {thr:2d}: parallel partition {row[0]:>7s} ms   serial partition {row[1]:>7s} ms", flush=True)
