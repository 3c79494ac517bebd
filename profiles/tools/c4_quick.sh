#!/bin/bash
# c4 (or $WL) lean bench line reduced to the figures used while tuning.  Usage: [WL=plate1m] bash profiles/tools/c4_quick.sh [ENV=VAL ...]
cd "${GRAFT_REPO_ROOT:-.}"
out=$(env "$@" timeout -k 10 200 python3 bench.py --workload ${WL:-c4} --lean --steps 5 --warmup 1 2>/dev/null) || { echo failed; exit 1; }
python3 - "$out" "$*" <<'PY'
import json,sys
d=json.loads(sys.argv[1]); r=d["roofline"]
print(f"[{sys.argv[2]}] step {d['ms_per_step']:.3f} ms  kernel {r['kernel_ms']:.3f} ms  value {d['value']/1e3:.2f} Grays/s  kernel-only {r['kernel_only_mrays_s']/1e3:.2f}")
PY
