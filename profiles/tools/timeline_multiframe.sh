#!/bin/bash
# Kernel timeline of the last frames of the bench's multi-frame job (8 meshes x 32768 tris, 1024x576, 50 spp, 2 cameras).
# Usage: bash profiles/tools/timeline_multiframe.sh [ENV=VAL ...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/trace_mf
for kv in "$@"; do export "$kv"; done
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_mf -- python3 $R/profiles/tools/multiframe.py 1 > $R/gpurun_out/trace_mf.log 2>&1 || { echo "trace failed"; tail -5 $R/gpurun_out/trace_mf.log; exit 1; }
tail -1 $R/gpurun_out/trace_mf.log
cd $R
python3 - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/trace_mf/*/*kernel_trace.csv")[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Start_Timestamp"]))
ren=[i for i,r in enumerate(rows) if "render_packet" in r["Kernel_Name"]]
i0=ren[-10]
t0=int(rows[i0]["Start_Timestamp"])
for r in rows[i0:]:
    s=(int(r["Start_Timestamp"])-t0)/1e6; e=(int(r["End_Timestamp"])-t0)/1e6
    name=r["Kernel_Name"].split("(")[0][-34:]
    if "jump_tree" in name: continue
    print("%8.3f %8.3f  %6.3f ms  q%s  %s  grid %s" % (s,e,e-s,r.get("Queue_Id","?"),name,r.get("Grid_Size_X",r.get("Grid_Size","?"))))
# busy fraction of the GPU over the last 60 render kernels
j0=ren[-60]; span=(int(rows[ren[-1]]["End_Timestamp"])-int(rows[j0]["Start_Timestamp"]))/1e6
ev=sorted((int(r["Start_Timestamp"]),int(r["End_Timestamp"])) for r in rows[j0:])
busy=0; cur_s,cur_e=ev[0]
for s,e in ev[1:]:
    if s>cur_e: busy+=cur_e-cur_s; cur_s,cur_e=s,e
    else: cur_e=max(cur_e,e)
busy+=cur_e-cur_s
print("last 60 frames: span %.1f ms, some kernel running %.1f ms (%.0f %%), render kernels %.1f ms" % (span, busy/1e6, 100*busy/1e6/span, sum((int(rows[i]["End_Timestamp"])-int(rows[i]["Start_Timestamp"])) for i in ren[-60:])/1e6))
PY
