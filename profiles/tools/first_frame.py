import sys, time
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "..", ".."))
import numpy as np, raytracer_amd as ra
from raytracer_amd import meshgen
c, k = meshgen.plate(128, side=5.0, x0=-2.5, y0=-2.5, z0=-6.0)
host = ra.HostScene([np.ascontiguousarray(c)], [np.ascontiguousarray(k)], 0)
ctx = ra.Context(0)
ctx.upload(host)
H, W, spp = 576, 1024, 50
cam = meshgen.camera_arrays(W, H)
for i in range(3):
    t0 = time.perf_counter(); ctx.render(*cam, H, W, spp); t1 = time.perf_counter(); ctx.sync(); t2 = time.perf_counter()
    s = ctx.stats()
    print("frame %d: render call %.2f ms, then sync %.2f ms; host_jump_ms %.2f, jitter kernel %.2f ms, traversal %.2f ms" % (i, (t1-t0)*1e3, (t2-t1)*1e3, s["host_jump_ms"], s["jitter_kernel_ms"], s["render_kernel_ms"]))
