#!/bin/bash
# Host BVH build time of the multi-frame scene (8 meshes x 32,768 triangles) and of one 262 K-triangle mesh against the
# pool width (RT_HOST_THREADS; one process each: the pool is created once).  Usage: bash profiles/tools/build_threads.sh
cd "${GRAFT_REPO_ROOT:-.}"
for t in 1 2 4 8 16; do
RT_HOST_THREADS=$t python3 - <<PY 2>&1 | tail -2
import os,sys,time
sys.path.insert(0,'.')
import numpy as np, raytracer_amd as ra
from raytracer_amd import meshgen
def scene(nm,n):
    cs,ks=[],[]
    for m in range(nm):
        c,k=meshgen.plate(n, side=5.0, x0=-10.0+5.0*(m%4), y0=-5.0+5.0*(m//4), z0=-6.0, T=1, sway=0.01)
        cs.append(np.ascontiguousarray(c)); ks.append(np.ascontiguousarray(k))
    return cs,ks
for nm,n in ((8,128),(1,362)):
    cs,ks=scene(nm,n)
    ts=[]
    for r in range(5):
        t0=time.time(); h=ra.HostScene(cs,ks,0); ts.append(time.time()-t0); h.close()
    print("RT_HOST_THREADS=$t", nm, "x", 2*n*n, "triangles: %.2f ms (min of 5)"%(min(ts)*1e3), flush=True)
PY
done
