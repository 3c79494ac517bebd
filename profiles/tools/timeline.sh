#!/bin/bash
# Kernel timeline of the last frames of a lean bench run (who overlaps whom): start / end in ms relative to the
# first listed kernel, stream id, name.  Usage: [WL=c4] [STEPS=3] bash profiles/tools/timeline.sh [ENV=VAL ...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/trace_tl
for kv in "$@"; do export "$kv"; done
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_tl -- python3 $R/bench.py --workload ${WL:-c4} --steps ${STEPS:-3} --warmup 1 --lean > /dev/null 2>&1 || { echo "trace failed"; exit 1; }
cd $R
python3 - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/trace_tl/*/*kernel_trace.csv")[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Start_Timestamp"]))
ren=[i for i,r in enumerate(rows) if "render_packet" in r["Kernel_Name"]]
# from the second render kernel to the last-but-one (the timed steps; the last two are the untimed extras)
i0=ren[1] if len(ren)>2 else 0
t0=int(rows[i0]["Start_Timestamp"])
for r in rows[i0:]:
    s=(int(r["Start_Timestamp"])-t0)/1e6; e=(int(r["End_Timestamp"])-t0)/1e6
    name=r["Kernel_Name"].split("(")[0][-40:]
    print("%9.3f %9.3f  %7.3f ms  q%s  %s  grid %s" % (s,e,e-s,r.get("Queue_Id","?"),name,r.get("Grid_Size_X",r.get("Grid_Size","?"))))
PY
