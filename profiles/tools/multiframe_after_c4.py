"""The multi-frame leg the way bench.py runs it: after the 5 M-triangle scene has been built and uploaded in the same
process.  Usage: python profiles/tools/multiframe_after_c4.py [repeats] [free]   ("free": drop the big scene first)"""
import gc
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402
import raytracer_amd as ra  # noqa: E402

label, coords, colors, cam, H, W, spp = bench.workload("c4")
host = ra.HostScene(coords, colors, 0)
ctx = ra.Context(0)
ctx.upload(host)
if len(sys.argv) > 2:
    ctx.close(); host.close(); del coords, colors, host, ctx; gc.collect()
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    r = bench.bench_multiframe(ra, 0, 1, 0, None)
    print("%.0f frames/s, CPU %.1f ms per timestep" % (r["frames_per_s"], r["cpu_ms_per_timestep"]), flush=True)
