"""Host BVH build time against thread count: 8 meshes x 32,768 triangles (the multi-frame bench scene) and one mesh of
262,088.  Usage: python profiles/tools/build_scaling.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import raytracer_amd as ra
from raytracer_amd import meshgen
def scene(n_mesh, n):
    cs, ks = [], []
    for m in range(n_mesh):
        c, k = meshgen.plate(n, side=5.0, x0=-10.0 + 5.0 * (m % 4), y0=-5.0 + 5.0 * (m // 4), z0=-6.0, T=1, sway=0.01)
        cs.append(np.ascontiguousarray(c)); ks.append(np.ascontiguousarray(k))
    return cs, ks
for (nm, n) in ((8,128),(1,362)):
    cs, ks = scene(nm, n)
    for thr in (1, 2, 4, 8, 16):
        ts=[]
        for t in range(3):
            t0=time.time(); h=ra.HostScene(cs, ks, 0, n_threads=thr); ts.append(time.time()-t0); h.close()
        print(nm, "meshes x", n, "threads", thr, "%.1f ms"%(min(ts)*1e3))
