"""Host supply of the multi-frame job's timestep scene (8 meshes x 32,768 triangles) in ONE process, built the way
rt_render_scene builds it (n_threads = -1: one thread per tree on the pool, several builds in flight).
Usage: RT_HOST_THREADS=<n> python profiles/tools/build_supply.py [in_flight [builds]]"""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np  # noqa: E402
import raytracer_amd as ra  # noqa: E402
from raytracer_amd import meshgen  # noqa: E402
from raytracer_amd import _lib as L  # noqa: E402

in_flight = int(sys.argv[1]) if len(sys.argv) > 1 else 6
n_builds = int(sys.argv[2]) if len(sys.argv) > 2 else 96
coords, colors = [], []
for m in range(8):
    c, k = meshgen.plate(128, side=5.0, x0=-10.0 + 5.0 * (m % 4), y0=-5.0 + 5.0 * (m // 4), z0=-6.0)
    coords.append(np.ascontiguousarray(c))
    colors.append(np.ascontiguousarray(k))


def build(_):
    ra.HostScene(coords, colors, 0, n_threads=-1).close()


with ThreadPoolExecutor(in_flight) as ex:
    list(ex.map(build, range(2 * in_flight)))
    cpu0, t0 = time.process_time(), time.perf_counter()
    list(ex.map(build, range(n_builds)))
    wall, cpu = time.perf_counter() - t0, time.process_time() - cpu0
print("pool threads %d, %d builds in flight: %.2f ms wall and %.1f ms CPU per timestep (%.0f timesteps/s, %.1f cores busy)" % (
    int(L.lib.rt_host_threads()), in_flight, wall / n_builds * 1e3, cpu / n_builds * 1e3, n_builds / wall, cpu / wall))
