#!/bin/bash
# Quick SQ-counter passes only (for kernel tuning).  Usage: bash profiles/run_sq.sh <tag> [workload]
set -o pipefail
TAG=${1:-sq}
WL=${2:-c4}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_${TAG}_${WL}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH1="python3 $REPO/bench.py --workload $WL --steps 1 --warmup 0 --cpu-rays 0"
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT/pmc_sq" -- $BENCH1 > "$OUT/pmc_sq.log" 2>&1 || { echo "sq pass failed"; tail -5 "$OUT/pmc_sq.log"; exit 1; }
echo "sq pass ok"
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq2" -- $BENCH1 > "$OUT/pmc_sq2.log" 2>&1 || { echo "sq2 pass failed"; tail -5 "$OUT/pmc_sq2.log"; exit 1; }
echo "sq2 pass ok"
