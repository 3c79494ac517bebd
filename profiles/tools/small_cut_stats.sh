#!/bin/bash
# per-kernel stats of the sphere bench with one cut level at the render tile (small-frame experiment)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
rm -rf $R/gpurun_out/trace_sc
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trace_sc -- python3 $R/bench.py --workload sphere --steps 40 --warmup 10 --lean > $R/gpurun_out/trace_sc.log 2>&1 || { echo failed; tail -5 $R/gpurun_out/trace_sc.log; exit 1; }
cut -d, -f1-4,6,7 $R/gpurun_out/trace_sc/*/*kernel_stats.csv | head -12
