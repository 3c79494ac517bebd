"""Does the learnt sampler stream count arrive, and does the step then equal the traversal kernel?  plate1m at a given
sample count for several (steps, warm-up) pairs.  Usage: python profiles/tools/adapt_check.py [spp]"""
import json
import os
import subprocess
import sys

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
spp = sys.argv[1] if len(sys.argv) > 1 else "64"
for steps, warm in ((5, 1), (20, 5), (40, 10)):
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "plate1m", "--steps", str(steps),
                          "--warmup", str(warm), "--lean"], env=dict(os.environ, RT_BENCH_SPP=spp), capture_output=True,
                         text=True, timeout=200).stdout
    d = json.loads(out.strip().splitlines()[-1])
    r = d["roofline"]
    print("spp %s, %d steps after %d: step %s ms, kernel %s ms (alone %s), sampler in the timed step %s ms, streams %s" % (
        spp, steps, warm, d["ms_per_step"], r["kernel_ms"], (r.get("kernel_alone") or {}).get("kernel_ms"),
        d["aux"].get("jitter_kernel_ms_timed_step"), d["aux"].get("jitter_streams")), flush=True)
