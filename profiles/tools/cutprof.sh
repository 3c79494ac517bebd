# VALU instructions and active lanes per instruction of each cut_level_kernel launch, with and without cut_trim (runs on the GPU box).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
for t in 0 1; do
  export RT_NO_TRIM=$t
  if [ $t = 0 ]; then unset RT_NO_TRIM; fi
  timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $R/gpurun_out/cutprof_$t -- python3 $R/bench.py --workload c4 --steps 1 --warmup 0 --lean > $R/gpurun_out/cutprof_$t.log 2>&1 || exit 1
done
python3 - <<PY
import csv,glob,collections
for t in (0,1):
    f=glob.glob("$R/gpurun_out/cutprof_%d/**/*counter_collection.csv"%t, recursive=True)[0]
    rows=list(csv.DictReader(open(f)))
    d=collections.defaultdict(dict)
    for r in rows:
        if 'cut_level' in r['Kernel_Name']:
            d[(r['Dispatch_Id'],r['Grid_Size'])][r['Counter_Name']]=float(r['Counter_Value'])
    print("NO_TRIM" if t else "TRIM")
    for k,v in sorted(d.items(), key=lambda x:int(x[0][0]))[:8]:
        iv=v.get('SQ_INSTS_VALU',0); tc=v.get('SQ_THREAD_CYCLES_VALU',0)
        print(k, 'VALU %.1fM'%(iv/1e6), 'lanes/instr %.1f'%(tc/iv/4 if iv else 0))
PY
