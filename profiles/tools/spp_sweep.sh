#!/bin/bash
# Kernel-only time of the plate1m scene at several sample counts / instantiations (tuning aid, GPU box).
# Usage: bash profiles/tools/spp_sweep.sh "<spp list>" [extra env assignments...]
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
SPPS=${1:-"50 64"}
shift
for s in $SPPS; do
  out=$(env RT_BENCH_SPP=$s "$@" timeout -k 10 120 python3 bench.py --workload plate1m --steps 20 --warmup 5 --lean 2>/dev/null) || { echo "spp $s failed"; exit 1; }
  python3 - "$s" "$out" <<'PY'
import json,sys
d=json.loads(sys.argv[2]); r=d["roofline"]
print(f"spp {sys.argv[1]:>4}: kernel {r['kernel_ms']:.3f} ms  {r['kernel_only_mrays_s']/1e3:.2f} Grays/s  step {d['ms_per_step']:.3f} ms  box tests/ray {r['hbm_per_wave']['box_tests_per_ray']:.2f}  tri tests/ray {r['hbm_per_wave']['tri_tests_per_ray']:.2f}  tri records/batch {r['hbm_per_wave']['tri_records_per_batch']:.2f}  necessary frac {r['frac']:.3f}")
PY
done
