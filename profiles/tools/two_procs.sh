#!/bin/bash
# Two processes sharing the one GPU (the rehearsal of a 2-rank job on a one-GPU box), each running the multi-frame job.
# Usage: bash profiles/tools/two_procs.sh [ENV=VAL ...]
cd "${GRAFT_REPO_ROOT:-.}"
for kv in "$@"; do export "$kv"; done
mkdir -p gpurun_out
LOCAL_WORLD_SIZE=2 LOCAL_RANK=0 timeout -k 5 70 python3 -u profiles/tools/multiframe.py 1 > gpurun_out/two_r0.txt 2>&1 &
p0=$!
LOCAL_WORLD_SIZE=2 LOCAL_RANK=1 timeout -k 5 70 python3 -u profiles/tools/multiframe.py 1 > gpurun_out/two_r1.txt 2>&1
wait $p0
echo "[$*] rank 0: $(tail -1 gpurun_out/two_r0.txt)"
echo "[$*] rank 1: $(tail -1 gpurun_out/two_r1.txt)"
