#!/bin/bash
# Which cut mode for frames of 1 M .. 12 M rays?  Step time of none / one level (breadth-first kernel, 31 or 12
# entries) / the whole hierarchy over image sizes, sample counts and scene sizes.  Usage: bash profiles/tools/cut_policy_sweep.sh
cd "${GRAFT_REPO_ROOT:-.}"
one() {  # workload size spp env...
  local wl=$1 size=$2 spp=$3; shift 3
  env RT_BENCH_SIZE=$size RT_BENCH_SPP=$spp "$@" timeout -k 10 120 python3 bench.py --workload $wl --steps 30 --warmup 8 --lean 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%.4f' % d['ms_per_step'])"
}
for wl in sphere plate32k plate1m; do
  for c in "400x225 16" "400x225 50" "400x225 128" "1024x576 2" "1024x576 8" "1024x576 16" "2048x2048 1" "2048x2048 2"; do
    set -- $c
    a=$(one $wl $1 $2 RT_CUT_MIN_RAYS=1e12 RT_CUT1_MIN_RAYS=1e12) || exit 1
    b=$(one $wl $1 $2 RT_CUT_MIN_RAYS=1e12 RT_CUT1_MIN_RAYS=0 RT_CUT1_CAP=31) || exit 1
    e=$(one $wl $1 $2 RT_CUT_MIN_RAYS=1e12 RT_CUT1_MIN_RAYS=0 RT_CUT1_CAP=12) || exit 1
    f=$(one $wl $1 $2 RT_CUT_MIN_RAYS=0) || exit 1
    echo "$wl $1 x $2 spp: none $a  one level/31 $b  one level/12 $e  hierarchy $f"
  done
done
