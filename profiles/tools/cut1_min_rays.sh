#!/bin/bash
# From how many rays does the one-level cut pay?  Sphere (configs[1] mesh, 400x225) at 1 .. 32 spp with and without.
cd "${GRAFT_REPO_ROOT:-.}"
for s in 1 4 8 16 24 32; do
  for m in 0 1e12; do
    RT_BENCH_SPP=$s RT_CUT1_MIN_RAYS=$m timeout -k 10 120 python3 bench.py --workload sphere --steps 40 --warmup 10 --lean 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('spp $s (%d rays) min_rays $m: step %.4f ms, kernel %.4f ms' % (d['roofline']['rays_per_launch'], d['ms_per_step'], r['kernel_ms']))" || exit 1
  done
done
