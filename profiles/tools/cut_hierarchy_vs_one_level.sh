#!/bin/bash
# small images at high sample counts (>= 12 M rays): the hierarchy against one breadth-first level
cd "${GRAFT_REPO_ROOT:-.}"
one() { local wl=$1 size=$2 spp=$3; shift 3
  env RT_BENCH_SIZE=$size RT_BENCH_SPP=$spp "$@" timeout -k 10 120 python3 bench.py --workload $wl --steps 30 --warmup 8 --lean 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%.4f' % d['ms_per_step'])"; }
for wl in sphere plate32k; do
  for c in "400x225 256" "400x225 512" "640x360 128" "800x450 64" "1024x576 50"; do
    set -- $c
    a=$(one $wl $1 $2) || exit 1
    b=$(one $wl $1 $2 RT_CUT_MIN_RAYS=1e12 RT_CUT1_MAX_TRIS_PER_BLOCK=1e9) || exit 1
    echo "$wl $1 x $2 spp: hierarchy $a  one level $b"
  done
done
