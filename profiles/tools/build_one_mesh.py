"""Single-thread build time of one 32,768-triangle mesh (the multi-frame job's unit of host work).
Usage: RT_HOST_THREADS=1 python profiles/tools/build_one_mesh.py"""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np, raytracer_amd as ra
from raytracer_amd import meshgen
c, k = meshgen.plate(128, side=5.0, x0=-10.0, y0=-5.0, z0=-6.0)
c = np.ascontiguousarray(c); k = np.ascontiguousarray(k)
ts = []
for _ in range(30):
    t0 = time.perf_counter(); h = ra.HostScene([c], [k], 0); ts.append(time.perf_counter() - t0); h.close()
print("one 32768-triangle mesh, RT_HOST_THREADS=%s: min %.2f ms median %.2f ms, nodes %d" % (os.environ.get("RT_HOST_THREADS"), min(ts)*1e3, sorted(ts)[15]*1e3, h.info.n_blas_nodes))
