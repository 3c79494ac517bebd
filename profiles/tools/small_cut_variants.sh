#!/bin/bash
# Mid-sized frames (2 M .. 12 M rays): one cut level at the render tile (cut_mode 1) against no cuts, its list
# length, with and without trimmed leaves, breadth-first kernel against the depth-first one, on the sphere bench
# (configs[1]) and the small-frame job.  Usage: bash profiles/tools/small_cut_variants.sh
cd "${GRAFT_REPO_ROOT:-.}"
run() {
  echo "== $*"
  env "$@" timeout -k 10 120 python3 bench.py --workload sphere --steps 40 --warmup 10 --lean 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('sphere: step %.4f ms, kernel %.4f ms, box tests/ray %.2f, tri tests/ray %.2f' % (d['ms_per_step'], r['kernel_ms'], r['hbm_per_wave']['box_tests_per_ray'], r['hbm_per_wave']['tri_tests_per_ray']))" || exit 1
  env "$@" timeout -k 10 200 python3 profiles/tools/small_frames.py 2>&1 | sed -n 2p || exit 1
}
run RT_CUT1_MIN_RAYS=1e12
run RT_CUT1_SERIAL=1
run RT_CUT1_CAP=8
run RT_CUT1_CAP=16
run RT_CUT1_CAP=24
run RT_CUT1_CAP=31
run RT_CUT1_CAP=16 RT_NO_TRIM=1
run RT_CUT1_CAP=31 RT_NO_TRIM=1
