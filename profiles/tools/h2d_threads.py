"""Pageable host-to-device copies: one thread against several (each with its own stream) on this box.
The runtime stages a pageable copy through its own pinned buffers in the calling thread, so one thread's rate
is a memcpy's, not the link's.  Usage: python profiles/tools/h2d_threads.py [MB]"""
import sys, time, threading
import torch
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 640
src = torch.empty(mb << 20, dtype=torch.uint8).fill_(1)
dst = torch.empty(mb << 20, dtype=torch.uint8, device="cuda")
pin = torch.empty(mb << 20, dtype=torch.uint8).pin_memory()
def copy(lo, hi, stream):
    with torch.cuda.stream(stream):
        dst[lo:hi].copy_(src[lo:hi], non_blocking=True)
        stream.synchronize()
for n in (1, 1, 2, 4, 8):
    streams = [torch.cuda.Stream() for _ in range(n)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step = (mb << 20) // n
    th = [threading.Thread(target=copy, args=(i * step, (i + 1) * step, streams[i])) for i in range(n)]
    [t.start() for t in th]; [t.join() for t in th]
    dt = time.perf_counter() - t0
    print(f"{n} threads: {dt * 1e3:.1f} ms, {mb / 1024 / dt:.1f} GiB/s")
torch.cuda.synchronize(); t0 = time.perf_counter(); dst.copy_(pin, non_blocking=True); torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"pinned, 1 thread: {dt * 1e3:.1f} ms, {mb / 1024 / dt:.1f} GiB/s")
