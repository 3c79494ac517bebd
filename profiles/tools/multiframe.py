"""The bench's multi-frame leg alone (8 meshes x 32,768 triangles rebuilt per timestep, 2 cameras, 1024x576, 50 spp),
several times.  Usage: python profiles/tools/multiframe.py [repeats [timesteps]]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402
import raytracer_amd as ra  # noqa: E402

for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    r = bench.bench_multiframe(ra, 0, 1, 0, None, steps_per_rank=int(sys.argv[2]) if len(sys.argv) > 2 else 48)
    print("%.0f frames/s (%d frames in %.1f ms), CPU %.1f ms per timestep, host-bound ceiling %.0f frames/s" % (
        r["frames_per_s"], r["frames"], r["seconds"] * 1e3, r["cpu_ms_per_timestep"],
        r["host_bound_ceiling"]["frames_per_s_per_rank"]), flush=True)
