#!/bin/bash
# Kernel timeline of the last frames of the reference's own kind of job (one 2048-triangle mesh, 400x225, 50 spp,
# 100 timesteps; the second, warm call of profiles/tools/cold_job.py).  Usage: bash profiles/tools/timeline_small.sh [ENV=VAL ...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
rm -rf $R/gpurun_out/trace_small
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_small -- python3 $R/profiles/tools/cold_job.py 100 32 > $R/gpurun_out/trace_small.log 2>&1 || { echo "trace failed"; tail -5 $R/gpurun_out/trace_small.log; exit 1; }
grep "warm" $R/gpurun_out/trace_small.log
cd $R
python3 - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/trace_small/*/*kernel_trace.csv")[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Start_Timestamp"]))
ren=[i for i,r in enumerate(rows) if "render_packet" in r["Kernel_Name"]]
i0=ren[-9]
t0=int(rows[i0]["Start_Timestamp"])
for r in rows[i0:]:
    s=(int(r["Start_Timestamp"])-t0)/1e6; e=(int(r["End_Timestamp"])-t0)/1e6
    print("%8.3f %8.3f  %6.3f ms  q%s  %s  grid %s" % (s,e,e-s,r.get("Queue_Id","?"),r["Kernel_Name"].split("(")[0][-34:],r.get("Grid_Size_X",r.get("Grid_Size","?"))))
j0=ren[-60]; span=(int(rows[ren[-1]]["End_Timestamp"])-int(rows[j0]["Start_Timestamp"]))/1e6
ev=sorted((int(r["Start_Timestamp"]),int(r["End_Timestamp"])) for r in rows[j0:])
busy=0; cs,ce=ev[0]
for s,e in ev[1:]:
    if s>ce: busy+=ce-cs; cs,ce=s,e
    else: ce=max(ce,e)
busy+=ce-cs
print("last 60 frames: span %.2f ms, some kernel running %.2f ms (%.0f %%), render kernels %.2f ms" % (span, busy/1e6, 100*busy/1e6/span, sum((int(rows[i]["End_Timestamp"])-int(rows[i]["Start_Timestamp"])) for i in ren[-60:])/1e6))
PY
