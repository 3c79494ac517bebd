import sys, time, os
sys.path.insert(0, '/root/repo')
import numpy as np, raytracer_amd as ra
from raytracer_amd import meshgen
cs, ks = meshgen.plate_grid(791, grid=(2, 2), side=10.0)
h = ra.HostScene(cs, ks, 0)
ctx = ra.Context(0)
for r in range(6):
    t0 = time.perf_counter(); ctx.upload(h); print("upload %.1f ms" % ((time.perf_counter() - t0) * 1e3))
