#!/bin/bash
# Kernel timeline of the last frames of the driver's "medium" case (32768 tris, 1024x576, 50 spp, 2 cameras).
# Usage: bash profiles/tools/timeline_driver.sh [ENV=VAL ...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/trace_drv
for kv in "$@"; do export "$kv"; done
cat > /tmp/drv_case.py <<PY
import sys
sys.path.insert(0, "$R/profiles/tools")
import small_frames as sf
sf.run_case("medium", 128, 24, 576, 1024, 50, 2)
PY
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_drv -- python3 /tmp/drv_case.py > $R/gpurun_out/trace_drv.log 2>&1 || { echo "trace failed"; tail -5 $R/gpurun_out/trace_drv.log; exit 1; }
tail -1 $R/gpurun_out/trace_drv.log
cd $R
python3 - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/trace_drv/*/*kernel_trace.csv")[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Start_Timestamp"]))
ren=[i for i,r in enumerate(rows) if "render_packet" in r["Kernel_Name"]]
i0=ren[-8]
t0=int(rows[i0]["Start_Timestamp"])
for r in rows[i0:]:
    s=(int(r["Start_Timestamp"])-t0)/1e6; e=(int(r["End_Timestamp"])-t0)/1e6
    name=r["Kernel_Name"].split("(")[0][-34:]
    if "rocclr" in name or "jump_tree" in name: continue
    print("%8.3f %8.3f  %6.3f ms  q%s  %s  grid %s" % (s,e,e-s,r.get("Queue_Id","?"),name,r.get("Grid_Size_X",r.get("Grid_Size","?"))))
PY
