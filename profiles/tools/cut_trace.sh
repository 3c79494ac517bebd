#!/bin/bash
# Per-level duration of the cut kernels of one frame + kernel totals (tuning aid, GPU box).  Usage: [WL=c4] bash profiles/tools/cut_trace.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/trace_cut
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trace_cut -- python3 $R/bench.py --workload ${WL:-c4} --steps 2 --warmup 1 --lean > /dev/null 2>&1 || { echo "trace failed"; exit 1; }
cd $R
python3 - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/trace_cut/*/*kernel_trace.csv")[0]
rows=list(csv.DictReader(open(f)))
cut=[r for r in rows if "cut_level" in r["Kernel_Name"]]
n=len(cut)//4 if len(cut)>=4 else len(cut)   # launches per frame (4 frames: warmup + 2 steps + 1 extra)
tot=0
for r in cut[-n:]:
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3; tot+=d
    print(r.get("Grid_Size_X", r.get("Grid_Size")), "%.1f us" % d)
print("cut levels total %.3f ms over %d launches" % (tot/1e3, n))
for r in csv.DictReader(open(glob.glob("gpurun_out/trace_cut/*/*kernel_stats.csv")[0])):
    print(r["Name"][:60], r["Calls"], "%.3f ms avg" % (float(r["AverageNs"])/1e6))
PY
