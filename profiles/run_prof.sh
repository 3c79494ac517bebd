#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace + the PMC passes the bench's roofline/traffic figures use.
# Usage: bash profiles/run_prof.sh <tag> [workload]
# Every pass is its own rocprofv3 run (no --pmc together with trace domains), each under a time limit.
set -o pipefail
TAG=${1:-r01}
WL=${2:-c4}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_${TAG}_${WL}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --workload $WL --steps 20 --warmup 5 --lean"   # the driver's step counts: the trace average and ms_per_step are comparable
BENCH1="python3 $REPO/bench.py --workload $WL --steps 1 --warmup 0 --lean"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/trace.log" 2>&1 || { echo "trace pass failed"; tail -5 "$OUT/trace.log"; exit 1; }
echo "trace pass ok"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $BENCH1 > "$OUT/pmc_fetch.log" 2>&1 || { echo "fetch pass failed"; tail -5 "$OUT/pmc_fetch.log"; exit 1; }
echo "fetch pass ok"
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $BENCH1 > "$OUT/pmc_write.log" 2>&1 || { echo "write pass failed"; tail -5 "$OUT/pmc_write.log"; exit 1; }
echo "write pass ok"
timeout -s KILL 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_l2" -- $BENCH1 > "$OUT/pmc_l2.log" 2>&1 || { echo "l2 pass failed"; tail -5 "$OUT/pmc_l2.log"; exit 1; }
echo "l2 pass ok"
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT/pmc_sq" -- $BENCH1 > "$OUT/pmc_sq.log" 2>&1 || { echo "sq pass failed"; tail -5 "$OUT/pmc_sq.log"; exit 1; }
echo "sq pass ok"
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq2" -- $BENCH1 > "$OUT/pmc_sq2.log" 2>&1 || { echo "sq2 pass failed"; tail -5 "$OUT/pmc_sq2.log"; exit 0; }
echo "sq2 pass ok"
# scalar path: SMEM instruction count / time, scalar data cache hits and misses (the traversal's node and triangle
# records are all scalar loads)
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_INST_LEVEL_SMEM --output-format csv -d "$OUT/pmc_sq3" -- $BENCH1 > "$OUT/pmc_sq3.log" 2>&1 || { echo "sq3 pass failed"; tail -5 "$OUT/pmc_sq3.log"; exit 0; }
echo "sq3 pass ok"
timeout -s KILL 300 rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_DATA_READ_REQ --output-format csv -d "$OUT/pmc_sqc" -- $BENCH1 > "$OUT/pmc_sqc.log" 2>&1 || { echo "sqc pass failed"; tail -5 "$OUT/pmc_sqc.log"; exit 0; }
echo "sqc pass ok"
python3 $REPO/profiles/summarize_pmc.py "$OUT" "$OUT/pmc_${WL}.json" > "$OUT/summarize.log" 2>&1 && echo "summary ok: $OUT/pmc_${WL}.json"
cp "$OUT"/trace/*/*kernel_stats.csv "$OUT/kernel_stats_${WL}.csv" 2>/dev/null || find "$OUT/trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats_${WL}.csv" \;
