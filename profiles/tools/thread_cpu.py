"""Per-thread CPU time of the multi-frame job (reads /proc/self/task after the run)."""
import os, sys
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "..", ".."))
import bench, raytracer_amd as ra
def snap():
    out = {}
    for tid in os.listdir("/proc/self/task"):
        try:
            f = open(f"/proc/self/task/{tid}/stat").read()
        except OSError:
            continue
        comm = f[f.index("(") + 1:f.rindex(")")]
        rest = f[f.rindex(")") + 2:].split()
        out[int(tid)] = (comm, (int(rest[11]) + int(rest[12])) * 10.0)  # ms (CLK_TCK 100)
    return out
bench.bench_multiframe(ra, 0, 1, 0, None)
a = snap()
r = bench.bench_multiframe(ra, 0, 1, 0, None, steps_per_rank=192)
b = snap()
print("%.0f frames/s, CPU %.1f ms per timestep" % (r["frames_per_s"], r["cpu_ms_per_timestep"]))
rows = sorted(((b[t][1] - a.get(t, (None, 0.0))[1], b[t][0], t) for t in b), reverse=True)
tot = sum(x[0] for x in rows)
print("threads alive at the end: %d, their CPU in the run: %.0f ms = %.1f ms per timestep" % (len(rows), tot, tot / 192))
for d, comm, t in rows[:30]:
    print("  %8.0f ms  %-16s tid %d%s" % (d, comm, t, "  (main)" if t == os.getpid() else ""))
