import sys, time
t00 = time.perf_counter()
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "..", ".."))
from raytracer_amd import _lib as L
t0 = time.perf_counter()
n = L.lib.rt_device_count()
t1 = time.perf_counter()
h = L.lib.rt_create(0)
t2 = time.perf_counter()
h2 = L.lib.rt_create(0)
t3 = time.perf_counter()
print("import %.1f ms, rt_device_count %.1f ms, first rt_create %.1f ms, second rt_create %.1f ms" % ((t0-t00)*1e3, (t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3))
