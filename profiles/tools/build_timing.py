import sys, time
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "..", ".."))
import numpy as np, raytracer_amd as ra
from raytracer_amd import meshgen
cs, ks = meshgen.plate_grid(791, grid=(2, 2), side=10.0)
for _ in range(3):
    t0 = time.time(); h = ra.HostScene(cs, ks, 0); print("total %.1f ms" % ((time.time() - t0) * 1e3), file=sys.stderr); h.close()
