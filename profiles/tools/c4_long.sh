#!/bin/bash
# Like c4_quick.sh with more steps (adaptive sampler stream count settles after a few frames).  Usage: [WL=c4] [STEPS=20] bash profiles/tools/c4_long.sh [ENV=VAL ...]
cd "${GRAFT_REPO_ROOT:-.}"
out=$(env "$@" timeout -k 10 300 python3 bench.py --workload ${WL:-c4} --lean --steps ${STEPS:-20} --warmup ${WARM:-3} 2>/dev/null) || { echo failed; exit 1; }
python3 - "$out" "$*" <<'PY'
import json,sys
d=json.loads(sys.argv[1]); r=d["roofline"]
print(f"[{sys.argv[2]}] step {d['ms_per_step']:.3f} ms  kernel {r['kernel_ms']:.3f} ms  value {d['value']/1e3:.2f} Grays/s  kernel-only {r['kernel_only_mrays_s']/1e3:.2f}")
PY
