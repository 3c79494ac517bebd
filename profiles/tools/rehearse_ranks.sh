#!/bin/bash
# Host-side supply for 1/2/4/8 ranks on THIS box, no GPU touched: bench.py --rehearse starts the ranks (gloo, 127.0.0.1),
# every rank builds the multi-frame driver's timestep scene on its own share of the cores, all ranks at once.
# Usage: bash profiles/tools/rehearse_ranks.sh [outfile]
OUT=${1:-/dev/stdout}
for n in 1 2 4 8; do
  timeout -k 10 300 python3 bench.py --gpus $n --rehearse 2>/dev/null | grep '^{' | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
h = d['host_build_supply']
b = [r['build'] for r in d['ranks']]
print('ranks %d: threads/rank %s, cores visible %d, per-rank wall %.1f ms and cpu %.1f ms per timestep, all ranks %.0f timesteps/s -> <= %.0f frames/s' % (
    d['n_gpus'], sorted(set(r['host_threads'] for r in d['ranks'])), h['cores_visible'],
    sum(x['wall_ms_per_timestep'] for x in b) / len(b), sum(x['cpu_ms_per_timestep'] for x in b) / len(b),
    h['timesteps_per_s_all_ranks'], h['frames_per_s_ceiling_2_cameras']))" >> $OUT
done
