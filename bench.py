#!/usr/bin/env python3
"""bench.py -- Mrays/s of the render hot path on MI355X (contract: see the round prompt / DESIGN.md).

A "step" is one full frame of the workload: MT19937 jitter generation (GPU GF(2) jump-ahead kernels +
jitter kernel, the host only supplies cached jump polynomials), the per-frame camera-relative records (prep
kernel) and cuts (cut kernels: the tree classified once per pixel block), ray generation,
TLAS/BLAS traversal, Moller-Trumbore, shading, in-order sample accumulation, quantisation and per-sample
hit-id output, with the flattened tree already resident in HBM (the host BVH build and the upload happen
before the timed region and are reported separately).  Nothing is carried over from one step to the next but
the sampler's look-ahead: records and cuts are made again for every frame, as a new camera or timestep would
need them.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c4|plate1m|sphere|block]

N > 1 is launched by torch.distributed.run, one rank per GPU; frames are sharded over ranks
(frame f = step*N + rank, each at its own position of the reference's global RNG stream), with no
data-path collective: only the barrier and the MAX-over-ranks of the elapsed time.
Prints ONE JSON line on rank 0.  Besides the contract's fields it carries, under "aux" and never as part of
`value`: the multi-frame driver's frames/s (BASELINE's second metric), the fp32 fast path with its measured
tolerance, and for N > 1 one frame split into row bands over the ranks (strong scaling) with its gather.
"""
import argparse
import socket
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
# fp64 VALU issue peak WITHOUT fma (-ffp-contract=off, the parity contract): 256 CU x 4 SIMD x 16 lanes x 2.4 GHz
# = 39.3 T lane-operations/s (MI355X_MICROARCH.md: 78.6 TFLOP/s fp64 vector counts an fma as two)
FP64_LANE_OPS_PEAK = 256 * 4 * 16 * 2.4e9
TDEV = "cuda"  # device of the tensors handed to torch.distributed (see RT_BENCH_BACKEND)


def kernel_source_hash():
    """sha256 of the device code the library carries (the .hip_fatbin section of raytracer_amd/librtgpu.so): the
    kernels' instruction streams themselves, unchanged by edits to host code or comments.  profiles/summarize_pmc.py
    stores it with every PMC summary; the bench refuses to price a launch with counters taken from other kernels."""
    import hashlib
    import struct
    b = open(os.path.join(ROOT, "raytracer_amd", "librtgpu.so"), "rb").read()
    if b[:4] != b"\x7fELF" or b[4] != 2:
        raise RuntimeError("librtgpu.so is not a 64-bit ELF file")
    shoff = struct.unpack_from("<Q", b, 0x28)[0]
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", b, 0x3A)

    def section(i):
        name, _typ, _flags, _addr, off, size = struct.unpack_from("<IIQQQQ", b, shoff + i * shentsize)
        return name, off, size

    stroff = section(shstrndx)[1]
    for i in range(shnum):
        name, off, size = section(i)
        if b[stroff + name:b.index(b"\0", stroff + name)] == b".hip_fatbin":
            return hashlib.sha256(b[off:off + size]).hexdigest()[:16]
    raise RuntimeError("librtgpu.so has no .hip_fatbin section")


def launch_ranks(n):
    """`--gpus N` without a launcher: start the N ranks ourselves as fresh child processes (this process has not
    touched the GPU and never will) and return their exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def workload(name):
    """-> (label, coords list, colors list, (cam, p00, spacing), H, W, spp)"""
    from raytracer_amd import meshgen
    if name == "c4":
        # BASELINE.json configs[3]: 5M-triangle deformed plate, 4096x4096, 64 spp, several BLAS under one TLAS
        n = 791  # 4 plates x 2*791^2 = 5,005,448 triangles
        coords, colors = meshgen.plate_grid(n, grid=(2, 2), side=10.0)
        H = W = 4096
        spp = 64
        label = "configs[3]: synthetic 5M-triangle deformed plate (4 BLAS / 1 TLAS), 4096x4096, 64 spp"
    elif name == "plate1m":
        c, k = meshgen.plate(724)  # 1,048,352 triangles, north_star's >=1M-triangle / 50 spp target
        coords, colors = [c], [k]
        H = W = 2048
        spp = 50
        label = "1,048,352-triangle deformed plate (1 BLAS), 2048x2048, 50 spp"
    elif name == "plate32k":  # (experiments: a mid-sized mesh, the multi-frame job's unit)
        c, k = meshgen.plate(128, side=5.0, x0=-2.5, y0=-2.5, z0=-4.0)
        coords, colors = [c], [k]
        H, W, spp = 576, 1024, 50
        label = "32,768-triangle deformed plate (1 BLAS), 1024x576, 50 spp"
    elif name == "sphere":
        # BASELINE.json configs[1]: the reference's sphere_gmsh_vtk_test_2d_manyelems.vtk (data fixture),
        # 1 camera, 50-sample anti-aliasing, 1 frame, at the reference's 400x225 (rttests.py:31-33)
        from raytracer_amd import rtsimdataloader as loader
        m = loader.vtk_mesh_to_rtmesh(os.path.join(ROOT, "tests", "golden", "sphere_gmsh_vtk_test_2d_manyelems.vtk"),
                                      scale=2.0, world_position=np.array([0.0, 0.0, -2.2]))
        m.set_surface(loader.SurfType.FIELD_COLOR, np.array([0.5, 0.5, 0.5]))
        coords, colors = [m.node_coords_expanded_over_time], [m.face_colors_over_time]
        H, W, spp = 225, 400, 50
        label = "configs[1]: sphere_gmsh_vtk_test_2d_manyelems.vtk (4,992 triangles), 400x225, 50 spp"
    elif name == "block":
        # BASELINE.json configs[0]: simple_block.e, single mesh, single camera, 1 frame, no anti-aliasing
        from raytracer_amd import rtsimdataloader as loader
        m = loader.simdata_to_rtmesh(os.path.join(ROOT, "tests", "golden", "simple_block.e"), scale=100.0,
                                     world_position=np.array([-0.5, -0.75, -2.0]), color_field="disp_x")
        coords, colors = [m.node_coords_expanded_over_time[4:5].copy()], [m.face_colors_over_time[4:5].copy()]
        H, W, spp = 225, 400, 1
        label = "configs[0]: simple_block.e surface (44 triangles), 400x225, 1 spp"
    else:
        raise SystemExit(f"unknown workload {name}")
    if os.environ.get("RT_BENCH_SPP"):  # experiments only: the same scene at another sample count
        spp = int(os.environ["RT_BENCH_SPP"])
        label += f" [RT_BENCH_SPP={spp}]"
    if os.environ.get("RT_BENCH_SIZE"):  # experiments only: WxH
        W, H = (int(v) for v in os.environ["RT_BENCH_SIZE"].lower().split("x"))
        label += f" [RT_BENCH_SIZE={W}x{H}]"
    cam = meshgen.camera_arrays(W, H)
    return label, coords, colors, cam, H, W, spp


def b_ray(nodes, tris, hits, rays, spp):
    """Algorithmic bytes per ray, SURVEY.md section 8(d): 64*N_node + 72*N_tri + 24*[hit] + 16 + 4 + 8/spp."""
    return 64.0 * nodes / rays + 72.0 * tris / rays + 24.0 * hits / rays + 16.0 + 4.0 + 8.0 / spp


def cpu_baseline(coords, colors, cam, H, W, spp, ctx_check, host, budget_rays):
    """The oracle (plain-C restatement, 1 thread) timed on a bounded, stratified sample of the same
    frame: whole rows spread over the image, each started at its exact position of the RNG stream.
    The same rows are compared with the GPU frame (hit ids bit-exact, intensity <= 1e-6)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc  # the checker; never on the product path
    import raytracer_amd as ra

    t0 = time.time()
    oscene = orc.Scene(coords, colors, 0)
    build_s = time.time() - t0
    n_rows = max(1, min(H, int(budget_rays // (W * spp))))
    rows = sorted(set(int((i + 0.5) * H / n_rows) for i in range(n_rows)))
    rays = 0
    secs = 0.0
    nodes = tris = hits = 0
    mismatches = 0
    max_abs = 0.0
    for r in rows:
        rng = orc.Rng()
        rng.set_state(ra.mt19937_state_at(r * W * spp * 4))
        t0 = time.time()
        w = oscene.render(*cam, H, W, spp, rng=rng, row0=r, row1=r + 1)
        secs += time.time() - t0
        rays += W * spp
        nodes += w["nodes"]
        tris += w["tris"]
        hits += w["hits"]
        if ctx_check is not None:
            g = ctx_check
            sl = slice(r * W * spp, (r + 1) * W * spp)
            mesh, elem = host.map_prims(g["hit_prim"][sl])
            mismatches += int((mesh != w["hit_mesh"]).sum() + (elem != w["hit_elem"]).sum())
            max_abs = max(max_abs, float(np.max(np.abs(g["intensity"][r] - w["intensity"][0]))))
    # the same oracle on every core this process may use (rows are independent once each starts at its
    # own position of the stream; ctypes releases the GIL): reported beside the 1-thread figure
    from concurrent.futures import ThreadPoolExecutor
    n_thr = max(1, min(16, len(os.sched_getaffinity(0))))
    mt_rows = (rows * ((4 * n_thr) // len(rows) + 1))[: max(len(rows), 4 * n_thr)] if n_thr > 1 else []
    states = {r: ra.mt19937_state_at(r * W * spp * 4) for r in set(mt_rows)}

    def one(r):
        g = orc.Rng()
        g.set_state(states[r])
        oscene.render(*cam, H, W, spp, rng=g, row0=r, row1=r + 1, want_hits=False)

    mt = None
    if mt_rows:
        t0 = time.time()
        with ThreadPoolExecutor(n_thr) as ex:
            list(ex.map(one, mt_rows))
        mt_s = time.time() - t0
        mt = dict(value=len(mt_rows) * W * spp / mt_s / 1e6, unit="Mrays/s", cores=n_thr,
                  sample=f"{len(mt_rows)} rows on {n_thr} threads, {mt_s:.1f} s")
    return dict(value=rays / secs / 1e6, unit="Mrays/s", cores=1, kind="port",
                sample=f"{len(rows)} full rows spread over the frame = {rays} rays, {secs:.1f} s "
                       f"(oracle BVH build {build_s:.1f} s not included)",
                all_cores=mt,
                nodes_per_ray=nodes / rays, tris_per_ray=tris / rays, hit_rate=hits / rays,
                b_ray=b_ray(nodes, tris, hits, rays, spp),
                gpu_vs_oracle=dict(hit_id_mismatches=mismatches, max_abs_intensity_diff=max_abs,
                                   rows_checked=len(rows)) if ctx_check is not None else None)


def bench_multiframe(ra, rank, world, local_rank, dist, steps_per_rank=48):
    """frames/s of rt_render_scene on a displacement sequence: 8 meshes x 32,768 triangles, 2 cameras,
    1024x576, 50 spp, `steps_per_rank * world` timesteps (each rank renders its share of the frames)."""
    import ctypes as C
    import tempfile
    from raytracer_amd import _lib as L, meshgen
    T = steps_per_rank * world
    H, W, spp = 576, 1024, 50
    coords, colors = [], []
    for m in range(8):
        c, k = meshgen.plate(128, side=5.0, x0=-10.0 + 5.0 * (m % 4), y0=-5.0 + 5.0 * (m // 4), z0=-6.0, T=T, sway=0.01)
        coords.append(np.ascontiguousarray(c))
        colors.append(np.ascontiguousarray(k))
    cams = [meshgen.camera_arrays(W, H, center=(-0.3, 0.0, 0.0), target=(0.0, 0.0, -6.0), fov=80),
            meshgen.camera_arrays(W, H, center=(0.3, 0.0, 0.0), target=(0.0, 0.0, -6.0), fov=80)]
    cc = np.ascontiguousarray(np.array([c[0] for c in cams], np.float64))
    p00 = np.ascontiguousarray(np.array([c[1] for c in cams], np.float64))
    sp = np.ascontiguousarray(np.array([c[2] for c in cams], np.float64).reshape(2, 6))
    cp = (C.c_void_p * 8)(*[c.ctypes.data for c in coords])
    kp = (C.c_void_p * 8)(*[k.ctypes.data for k in colors])
    counts = np.array([k.shape[1] for k in colors], np.int64)
    out_dir = tempfile.mkdtemp(prefix=f"rtbench_r{rank}_")

    def run(n_steps):
        # the driver prints the reference's progress lines: keep them off our one-JSON-line stdout
        sys.stdout.flush()
        saved = os.dup(1)
        devnull = os.open(os.devnull, os.O_WRONLY)
        os.dup2(devnull, 1)
        try:
            rc = L.lib.rt_render_scene(H, W, spp, out_dir.encode(), n_steps, 8, cp, kp, L.ptr(counts), 2, L.ptr(cc),
                                       L.ptr(p00), L.ptr(sp), local_rank, rank, world)
        finally:
            os.dup2(saved, 1)
            os.close(saved)
            os.close(devnull)
        L.check(rc, "rt_render_scene")

    # warm-up: polynomials, device buffers, and the host blocks of every build the driver keeps in flight (six
    # timesteps x 8 meshes per rank: a two-timestep warm-up left most of them to be page-faulted in the timed run)
    run(min(T, 12 * world))
    if dist is not None:
        dist.barrier()
    cpu0 = time.process_time()  # CPU time of this process, all threads (build pool, helpers, file writer, HIP waits)
    t0 = time.perf_counter()
    run(T)
    elapsed = time.perf_counter() - t0
    cpu_s = time.process_time() - cpu0
    my_steps = len(set(int(f) // 2 for f in ra.shard_frames(T, 2, rank, world)[0]))
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device=TDEV)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    n_files = len([f for f in os.listdir(out_dir) if f.endswith(".ppm")])
    for f in os.listdir(out_dir):
        os.remove(os.path.join(out_dir, f))
    os.rmdir(out_dir)
    frames = T * 2
    threads = int(L.lib.rt_host_threads())
    cpu_ms_step = cpu_s / max(1, my_steps) * 1e3
    return {"frames_per_s": frames / elapsed, "frames": frames, "seconds": elapsed, "image": [H, W], "spp": spp,
            # what the HOST can supply: this rank spent cpu_ms_per_timestep of CPU time per timestep it rendered (tree
            # builds, staging, read-back waits, P6 files) and has `host_threads` cores of its own, so however fast the
            # GPU is it cannot render more than host_threads / cpu time per timestep -- per rank; x world for the job
            "cpu_ms_per_timestep": cpu_ms_step,
            "host_bound_ceiling": {"host_threads_per_rank": threads,
                                   "frames_per_s_per_rank": threads / (cpu_ms_step * 1e-3) * 2 if cpu_ms_step > 0 else None,
                                   "frames_per_s_job": world * threads / (cpu_ms_step * 1e-3) * 2 if cpu_ms_step > 0 else None,
                                   "note": "cores of this rank / CPU time per timestep x 2 cameras (rank 0's figures)"},
            "meshes": 8, "triangles": int(counts.sum()), "cameras": 2, "timesteps": T,
            "mrays_per_s": frames * H * W * spp / elapsed / 1e6, "files_written_by_rank0": n_files,
            "includes": "per-timestep host BVH build (overlapped), upload, render, read-back, P6 write"}


def rehearse(rank, world, local_rank):
    """The N > 1 launch path on CPUs: rendezvous (gloo, 127.0.0.1), every rank's host thread count and its share of
    a frame sequence and of one banded frame, gathered on rank 0 and printed as one JSON line.  No GPU call."""
    import torch.distributed as dist
    import raytracer_amd as ra
    from raytracer_amd import _lib as L
    if "RANK" not in os.environ:  # --gpus 1 without a launcher: a world of one
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo")
    H, W, spp, T, n_cam = 4096, 4096, 64, 100, 2
    frames, pos = ra.shard_frames(T, n_cam, rank, world, 0, H * W * spp * 4)
    mine = {"rank": rank, "local_rank": local_rank, "host_threads": int(L.lib.rt_host_threads()),
            "omp_num_threads_env": os.environ.get("OMP_NUM_THREADS"), "frames": [int(f) for f in frames],
            "band": list(ra.shard_rows(H, W, spp, rank, world, 0))[:2]}
    # what this node's HOST can supply to `world` ranks: every rank builds the multi-frame driver's timestep scene
    # (8 meshes x 32,768 triangles, rebuilt per timestep) on its own share of the cores, all ranks at once -- no GPU.
    if os.environ.get("RT_REHEARSE_BUILDS", "1") != "0":
        from raytracer_amd import meshgen
        coords, colors = [], []
        for m in range(8):
            c, k = meshgen.plate(128, side=5.0, x0=-10.0 + 5.0 * (m % 4), y0=-5.0 + 5.0 * (m // 4), z0=-6.0)
            coords.append(np.ascontiguousarray(c))
            colors.append(np.ascontiguousarray(k))
        # as rt_render_scene builds them: six timesteps in flight, every tree by one thread of the rank's pool
        from concurrent.futures import ThreadPoolExecutor
        in_flight, n_builds = 6, 48

        def build(_):
            ra.HostScene(coords, colors, 0, n_threads=-1).close()  # (ctypes releases the GIL for the call)

        with ThreadPoolExecutor(in_flight) as ex:
            list(ex.map(build, range(in_flight)))  # warm-up: the pool's threads, the recycled blocks
            dist.barrier()
            cpu0, t0 = time.process_time(), time.perf_counter()
            list(ex.map(build, range(n_builds)))
            wall, cpu = time.perf_counter() - t0, time.process_time() - cpu0
        mine["build"] = {"timesteps_per_s": n_builds / wall, "wall_ms_per_timestep": wall / n_builds * 1e3,
                         "cpu_ms_per_timestep": cpu / n_builds * 1e3}
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        out = {"rehearsal": True, "n_gpus": world, "ranks": everyone}
        if all("build" in r for r in everyone):
            tps = sum(r["build"]["timesteps_per_s"] for r in everyone)
            out["host_build_supply"] = {
                "scene": "8 meshes x 32,768 triangles per timestep (bench_multiframe's)", "cores_visible": len(os.sched_getaffinity(0)),
                "timesteps_per_s_all_ranks": tps, "frames_per_s_ceiling_2_cameras": 2 * tps,
                "builds_in_flight_per_rank": 6,
                "note": "tree builds only (no staging, read-back, files): an upper bound on the job's frames/s that no "
                        "number of GPUs lifts; compare across --gpus 1/2/4/8 on the same host"}
        print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--workload", default=os.environ.get("RT_BENCH_WORKLOAD", "c4"))
    ap.add_argument("--cpu-rays", type=float, default=40e6, help="rays in the CPU-baseline sample, about 10 s of one core (0 = skip)")
    ap.add_argument("--no-hit-ids", action="store_true")
    ap.add_argument("--no-multiframe", action="store_true", help="skip the frames/s (multi-frame driver) measurement")
    ap.add_argument("--lean", action="store_true",
                    help="timed loop + roofline only (no CPU baseline, driver legs, fp32 / Philox legs): profiling runs")
    ap.add_argument("--rehearse", action="store_true",
                    help="N > 1 launch path without a GPU: ranks rendezvous over gloo, report their shard schedules and "
                         "host thread counts, and exit (CPU test of the launcher)")
    args = ap.parse_args()
    if args.lean:
        args.cpu_rays = 0
        args.no_multiframe = True

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # not under a launcher: become one.  Nothing above has touched the GPU (numpy only), the ranks are
        # fresh child processes, and this process exits with their code.
        raise SystemExit(launch_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with `python bench.py --gpus {args.gpus}` "
                         f"(starts the ranks itself) or `python -m torch.distributed.run --nproc-per-node {args.gpus} "
                         f"bench.py --gpus {args.gpus}`")
    if args.rehearse:
        return rehearse(rank, world, local_rank)
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        # RT_BENCH_BACKEND=gloo: rehearsal of the N > 1 logic on a box with fewer GPUs than ranks (ranks share
        # GPUs, collectives on the host); the driver's runs use the default, RCCL
        global TDEV
        if os.environ.get("RT_BENCH_BACKEND", "nccl") == "gloo":
            local_rank = local_rank % max(1, torch.cuda.device_count())
            torch.cuda.set_device(local_rank)
            dist.init_process_group("gloo")
            TDEV = "cpu"
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import raytracer_amd as ra

    label, coords, colors, cam, H, W, spp = workload(args.workload)
    t0 = time.time()
    host = ra.HostScene(coords, colors, 0)
    build_s = time.time() - t0
    ctx = ra.Context(local_rank)
    t0 = time.time()
    ctx.upload(host)
    upload_s = time.time() - t0
    rays_per_frame = H * W * spp
    frame_words = rays_per_frame * 4
    want_ids = not args.no_hit_ids

    def frame(step_idx, **kw):
        # frame f of the sequence sits at position f*frame_words of the reference's global stream
        f = step_idx * world + rank
        ctx.rng_seek(f * frame_words)
        ctx.render(*cam, H, W, spp, hit_ids=want_ids, **kw)

    def barrier():
        ctx.sync()
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()

    for s in range(args.warmup):
        frame(s)
    barrier()
    # kernel time of exactly the timed launches: the library adds every launch's HIP-event time (events on the
    # kernel's own stream) to a running total when the NEXT rt_render finds it complete, so no per-step sync
    before = ctx.stats()
    t0 = time.perf_counter()
    for s in range(args.steps):
        frame(args.warmup + s)
    barrier()
    elapsed = time.perf_counter() - t0
    after = ctx.stats()
    timed_launches = after["render_launches_timed"] - before["render_launches_timed"]
    kernel_ms_timed = ((after["render_kernel_ms_total"] - before["render_kernel_ms_total"]) / timed_launches
                       if timed_launches == args.steps else None)
    jitter_ms_last = after["jitter_kernel_ms"]
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device=TDEV)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # second metric of BASELINE.json ("frames/s multi-frame render"): the whole reference loop through
    # rt_render_scene -- per-timestep host BVH rebuild (TLAS + 8 BLAS), upload, 2 cameras x 50 spp,
    # read-back, P6 files -- frames sharded over ranks, no collective.  configs[4]-shaped, weak scaling.
    multiframe = None if args.no_multiframe else bench_multiframe(ra, rank, world, local_rank, dist)

    # N > 1 only: ONE frame of the same workload split into row bands over the ranks (SURVEY 8(e), strong
    # scaling): every rank renders its band at its own stream position, then the bands are gathered on rank 0
    # (RCCL gather of the f64 intensity bands).  Reported in aux, never part of `value`.  A rank that fails
    # must not leave its peers blocked in a collective: the ranks first agree (all_reduce of an error flag)
    # that every local phase succeeded, and only then enter the gather.
    banded = None
    if world > 1:
        import torch
        r0, r1, pos = ra.shard_rows(H, W, spp, rank, world, 0)
        sizes = [ra.shard_rows(H, W, spp, r, world, 0) for r in range(world)]
        err, t_render, band = "", 0.0, None
        try:
            def band_step():
                if r1 > r0:
                    ctx.rng_seek(pos)
                    ctx.render(*cam, H, W, spp, rows=(r0, r1), hit_ids=want_ids)

            band_step()
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                band_step()
            ctx.sync()
            t_render = (time.perf_counter() - t0) / args.steps
            t0 = time.perf_counter()
            band = ctx.download(rgb=False, hit_ids=False)["intensity"] if r1 > r0 else np.empty((0, W))
        except Exception as e:  # local work only: no collective inside this block
            err = f"rank {rank}: {e}"
        flag = torch.tensor([1.0 if err else 0.0], dtype=torch.float64, device=TDEV)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if float(flag.item()) > 0.0:
            banded = {"error": err or "another rank failed in the banded leg"}
        else:
            pad = max(b - a for a, b, _ in sizes)
            mine = torch.zeros(pad, W, dtype=torch.float64, device=TDEV)
            mine[: r1 - r0] = torch.from_numpy(np.ascontiguousarray(band)).to(TDEV)
            parts = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
            dist.gather(mine, parts, dst=0)
            torch.cuda.synchronize()
            t_gather = time.perf_counter() - t0
            tt = torch.tensor([t_render, t_gather], dtype=torch.float64, device=TDEV)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            banded = {"ms_per_frame_render": float(tt[0]) * 1e3, "mrays_per_s": rays_per_frame / float(tt[0]) / 1e6,
                      "download_plus_gather_ms": float(tt[1]) * 1e3, "bands": [[a, b] for a, b, _ in sizes],
                      "scaling": "strong"}

    # untimed extras on rank 0: visit counters, roofline, hand-over costs, CPU baseline
    out = None
    if rank == 0:
        frame(0)
        ctx.sync()
        st1 = ctx.stats()
        jms, hjm = st1["jitter_kernel_ms"], st1["host_jump_ms"]
        # kernel time: the timed steps' own launches (HIP events on the kernel's stream)
        kms = kernel_ms_timed if kernel_ms_timed else st1["render_kernel_ms"]
        # visit counters, twice: (a) the walk the TIMED kernel does -- from the pixel blocks' cuts, DESIGN 4.2a -- which
        # is what the roofline below prices; (b) the reference's visit rule (walk from node 0, every ray on its own),
        # which is what SURVEY 8(d)'s B_ray is defined on and is kept only under `reference_visit_rule`
        frame(0, counters=True)
        ctx.sync()
        ref_cs = ctx.stats()
        os.environ["RT_COUNT_WITH_CUTS"] = "1"
        try:
            frame(0, counters=True)
            ctx.sync()
            cs = ctx.stats()
        finally:
            os.environ.pop("RT_COUNT_WITH_CUTS", None)
        rays = cs["rays"]
        bray = b_ray(ref_cs["nodes_visited"], ref_cs["tris_tested"], ref_cs["rays_hit"], rays, spp)
        reference_visit_rule = {
            "nodes_per_ray": ref_cs["nodes_visited"] / rays, "tris_per_ray": ref_cs["tris_tested"] / rays,
            "b_ray_bytes": bray, "bytes_per_launch": bray * rays_per_frame,
            "gbs_at_kernel_time": rays_per_frame * bray / (kms * 1e-3) / 1e9,
            "x_hbm_peak": rays_per_frame * bray / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "note": "SURVEY 8(d)'s B_ray charges every RAY 64 B per node and 72 B per triangle it visits when it walks "
                    "from node 0 alone.  The timed kernel does not do that work (one scalar fetch per 64-ray wave, walk "
                    "started at the block's cut), so this is a model of the reference's visit rule and NOT a bound on "
                    "the kernel: x_hbm_peak > 1 by design.  Counted by a separate root-walk launch."}
        # bytes the timed kernel's waves really request (cut walk): 64 B per node record, 128 B per camera-relative
        # triangle record, 128 B per cut list + the per-ray streams (jitter in, hit id out, colour, intensity)
        wave_bytes = (cs["wave_node_fetches"] * 64.0 + cs["wave_tri_fetches"] * 128.0 + (rays / 64.0) * 128.0 +
                      rays * (16.0 + (4.0 if want_ids else 0.0) + 8.0 / spp) + cs["rays_hit"] * 24.0)
        hbm_wave = {"bytes_per_launch": wave_bytes, "achieved_gbs": wave_bytes / (kms * 1e-3) / 1e9,
                    "peak_gbs": HBM_PEAK_GBS, "frac": wave_bytes / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "node_records_per_batch": cs["wave_node_fetches"] / (rays / 64.0),
                    "tri_records_per_batch": cs["wave_tri_fetches"] / (rays / 64.0),
                    "box_tests_per_ray": cs["nodes_visited"] / rays, "tri_tests_per_ray": cs["tris_tested"] / rays,
                    "note": "counted on the cut walk (RT_COUNT_WITH_CUTS=1), i.e. by the walk the timed kernel does; "
                            "requested by the waves -- scalar cache / L2 / MALL absorb part of it, `traffic` is what "
                            "reached HBM"}
        # ALGORITHMIC work of a launch in fp64 lane-operations: what the frame needs under the cut walk, priced at the
        # leanest correctly rounded instruction sequences (DESIGN 4.2 "necessary work"), independent of how the kernel
        # is written: masked-off lanes, scalar bookkeeping moved to the vector unit, spills etc. are NOT in it.
        nec_parts = {
            # 2 int->f64, 2 adds (sample position), 12 pixel sample, 3 dd, 5 squared norm, 10 sqrt (v_rsq + Goldschmidt),
            # 5 refined reciprocal of the norm + 3 x 3 quotients, 3 x 7 reciprocals of the direction (slab test)
            "ray_generation": 69.0 * rays,
            "box_tests": 12.0 * cs["nodes_visited"],  # 6 mul + 3 max + 2 min + 1 cmp (octant-specialised slab test)
            "triangle_tests": 28.0 * cs["tris_tested"],  # camera-relative record: p, det, u/v numerators, rejects
            "shading": 17.0 * cs["rays_hit"] + 12.0 * (rays - cs["rays_hit"]),  # w + 3 x (3 mul + 2 add) | sky
            "in_order_sum": 3.0 * rays + 16.0 * (rays / spp)}  # 3 adds per sample; luma, /spp, *255.99, cvt per pixel
        necessary = sum(nec_parts.values())
        alone_ms = None
        try:
            if args.lean:  # profiling runs: the trace should hold the timed launches and little else
                raise StopIteration
            os.environ["RT_NO_LOOKAHEAD"] = "1"
            os.environ["RT_NO_PREP_OVERLAP"] = "1"
            al = []
            for _ in range(3):
                frame(0)
                ctx.sync()
                al.append(ctx.stats()["render_kernel_ms"])
            alone_ms = float(np.mean(al[1:]))
        except StopIteration:
            pass
        finally:
            os.environ.pop("RT_NO_LOOKAHEAD", None)
            os.environ.pop("RT_NO_PREP_OVERLAP", None)
        traffic = None
        lane_ops = None
        active_lane_ops = None
        step_ops = None
        step_kernels = None
        scalar_instr = None
        pmc_note = None
        pmc_path = os.path.join(ROOT, "profiles", f"pmc_{args.workload}.json")
        if os.path.exists(pmc_path):
            try:
                pmc = json.load(open(pmc_path))
                if pmc.get("kernel_source_hash") != kernel_source_hash():
                    pmc_note = (f"profiles/pmc_{args.workload}.json was collected on other kernel sources "
                                f"({pmc.get('kernel_source_hash')} != {kernel_source_hash()}): re-run profiles/run_prof.sh")
                    print("bench.py: WARNING: " + pmc_note, file=sys.stderr)
                elif pmc.get("spp") not in (None, spp) or pmc.get("image") not in (None, [H, W]):
                    pmc_note = f"profiles/pmc_{args.workload}.json is for another frame shape"
                else:
                    traffic = pmc.get("hbm_bytes_per_launch")
                    dom = pmc.get(pmc.get("dominant_kernel", ""), {})
                    vi = dom.get("SQ_INSTS_VALU")
                    lane_ops = vi * 64 if vi else None
                    # SQ_THREAD_CYCLES_VALU = sum over VALU instructions of their ACTIVE lanes (x 4 cycles on the 16-lane
                    # SIMD is already folded out by rocprofv3's unit: it reports lane-instructions): lanes under EXEC
                    active_lane_ops = dom.get("SQ_THREAD_CYCLES_VALU")
                    if all(k in dom for k in ("SQ_INSTS_SALU", "SQ_INSTS_BRANCH", "SQ_INSTS_SMEM")):
                        scalar_instr = dom["SQ_INSTS_SALU"] + dom["SQ_INSTS_BRANCH"] + dom["SQ_INSTS_SMEM"]
                    # every kernel of a steady-state step (they share the GPU with each other): launches per step x
                    # SQ_INSTS_VALU per launch
                    per_step = {"render_packet_kernel": 1, "mt_jitter_kernel": 1, "mt_jump_all_kernel": 1,
                                "mt_jump_kernel": 1, "prep_frame_kernel": 1,
                                "cut_level_kernel": 1}  # (the summary keeps the finest level's launch; the coarser one is ~1/16 of it)
                    step_kernels = {}
                    for name, rec in pmc.items():
                        if not isinstance(rec, dict) or "SQ_INSTS_VALU" not in rec or "<true" in name:
                            continue
                        for key, mult in per_step.items():
                            if key in name:
                                step_kernels[key] = rec["SQ_INSTS_VALU"] * 64.0 * mult
                    step_ops = sum(step_kernels.values()) if step_kernels else None
                    pmc_note = (f"profiles/pmc_{args.workload}.json (SQ_INSTS_VALU x 64, SQ_THREAD_CYCLES_VALU; "
                                f"FETCH_SIZE/WRITE_SIZE passes)")
            except Exception as e:
                pmc_note = f"profiles/pmc_{args.workload}.json unreadable: {e}"
        else:
            pmc_note = f"no profiles/pmc_{args.workload}.json: run profiles/run_prof.sh <tag> {args.workload}"
        peak_t = FP64_LANE_OPS_PEAK / 1e12
        achieved = necessary / (kms * 1e-3) / 1e12

        def rate(ops, ms):
            return ops / (ms * 1e-3) / 1e12 if ops and ms else None

        def frac(ops, ms):
            return ops / (ms * 1e-3) / FP64_LANE_OPS_PEAK if ops and ms else None

        roofline = {"bound": "fp64_valu_issue", "achieved": achieved, "peak": peak_t, "unit": "T lane-ops/s",
                    "frac": achieved / peak_t, "traffic": traffic,
                    "what": "achieved = ALGORITHMIC fp64 lane-operations of one launch (`necessary`, below) / the kernel's "
                            "average duration in the timed steps; frac = distance from the work the frame needs.  The "
                            "instructions the kernel really issues are under `issue` (a busy-ness figure: it rises when "
                            "instructions are added)",
                    "kernel": "rtk::render_packet_kernel", "kernel_ms": kms,
                    "kernel_ms_source": ("HIP events of the %d timed launches" % args.steps) if kernel_ms_timed
                    else "HIP events of one launch after the timed loop",
                    "necessary": {"lane_ops_per_launch": necessary, "lane_ops_per_ray": necessary / rays,
                                  "by_part": nec_parts,
                                  "box_tests_per_ray": cs["nodes_visited"] / rays,
                                  "tri_tests_per_ray": cs["tris_tested"] / rays, "hit_rate": cs["rays_hit"] / rays,
                                  "counts_from": "RT_COUNT_WITH_CUTS=1 launch of this run (the timed kernel's own walk)"},
                    "efficiency": necessary / lane_ops if lane_ops else None,  # necessary / executed (x64)
                    "efficiency_vs_active_lanes": necessary / active_lane_ops if active_lane_ops else None,
                    "issue": {"executed_lane_ops_x64": lane_ops, "active_lane_ops": active_lane_ops,
                              "active_lanes_per_instruction": active_lane_ops / (lane_ops / 64.0)
                              if lane_ops and active_lane_ops else None,
                              "frac_x64": frac(lane_ops, kms), "frac_active_lanes": frac(active_lane_ops, kms),
                              "counters_from": pmc_note,
                              "note": "x64 counts every issue slot, masked-off lanes included (port busy-ness); "
                                      "active lanes = SQ_THREAD_CYCLES_VALU (lanes under EXEC)"},
                    "kernel_alone": {"kernel_ms": alone_ms, "achieved": rate(necessary, alone_ms),
                                     "frac": frac(necessary, alone_ms), "issue_frac_x64": frac(lane_ops, alone_ms),
                                     "issue_frac_active_lanes": frac(active_lane_ops, alone_ms),
                                     "note": "the same kernel with the GPU to itself (no look-ahead sampler, prep / cut "
                                             "kernels in front of it instead of beside the previous launch)"},
                    # the other issue port: SALU + branch + SMEM instructions of the kernel against one instruction per
                    # SIMD per 4 cycles (1024 SIMDs x 2.4 GHz / 4)
                    "scalar_issue": {"instructions_per_launch": scalar_instr, "peak_ginstr_s": 1024 * 2.4 / 4,
                                     "frac_alone": scalar_instr / (alone_ms * 1e-3) / (1024 * 2.4e9 / 4)
                                     if scalar_instr and alone_ms else None,
                                     "frac": scalar_instr / (kms * 1e-3) / (1024 * 2.4e9 / 4) if scalar_instr else None},
                    "gpu_step": {"lane_ops_per_step_x64": step_ops, "by_kernel": step_kernels,
                                 "ms_per_step": elapsed / args.steps * 1e3,
                                 "issue_frac_x64": step_ops / (elapsed / args.steps) / FP64_LANE_OPS_PEAK if step_ops else None,
                                 "necessary_frac": necessary / (elapsed / args.steps) / FP64_LANE_OPS_PEAK,
                                 "note": "every kernel of a step (traversal, sampler, jumps, prep, cuts) against the step's "
                                         "wall time; necessary_frac = the frame's algorithmic work / step time / peak"},
                    "peak_note": "256 CU x 4 SIMD x 16 lanes x 2.4 GHz, no fma (-ffp-contract=off is the parity contract)",
                    "rays_per_launch": rays_per_frame, "kernel_only_mrays_s": rays_per_frame / (kms * 1e-3) / 1e6,
                    "hbm_measured_frac": traffic / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS if traffic else None,
                    "hbm_per_wave": hbm_wave, "reference_visit_rule": reference_visit_rule}
        # host hand-over cost (PCIe, page-locked host buffers), never part of `value`
        frame(0)
        ctx.sync()
        t1 = time.perf_counter()
        ctx.download(hit_ids=False)
        dl_frame_ms = (time.perf_counter() - t1) * 1e3
        dl_ids_ms = None
        if want_ids:
            t1 = time.perf_counter()
            ctx.download(intensity=False, rgb=False, hit_ids=True)
            dl_ids_ms = (time.perf_counter() - t1) * 1e3
        # SURVEY 8(d)'s wall metric for this config: per frame, host BVH build + upload + render + read-back
        end_to_end = None
        if not args.lean:
            try:
                e2e = []
                for i in range(3):
                    t1 = time.perf_counter()
                    h2 = ra.HostScene(coords, colors, 0)
                    ctx.upload(h2)
                    ctx.rng_seek(0)
                    ctx.render(*cam, H, W, spp, hit_ids=want_ids)
                    ctx.download(hit_ids=False)
                    t2 = time.perf_counter()
                    if want_ids:
                        ctx.download(intensity=False, rgb=False, hit_ids=True)
                    t3 = time.perf_counter()
                    h2.close()
                    e2e.append((t2 - t1, t3 - t1))
                fr = float(np.mean([x[0] for x in e2e[1:]]))
                fi = float(np.mean([x[1] for x in e2e[1:]]))
                end_to_end = {"ms_per_frame": fr * 1e3, "mrays_per_s": rays_per_frame / fr / 1e6,
                              "ms_per_frame_with_hit_ids": fi * 1e3 if want_ids else None,
                              "mrays_per_s_with_hit_ids": rays_per_frame / fi / 1e6 if want_ids else None,
                              "includes": "host BVH build (rt_scene_build, %d threads) + upload + sampler + render + "
                                          "read-back of intensity and rgb into page-locked host memory; serial, "
                                          "nothing overlapped" % int(ra.lib.rt_host_threads())}
            except Exception as e:
                end_to_end = {"error": str(e)}
        # the whole driver on big frames: 4096^2 x 64 spp, 8 meshes rebuilt per timestep, 2 cameras, P6 files
        soak = None
        if not args.lean and not args.no_multiframe and world == 1 and H * W >= 4096 * 4096:
            try:
                import tempfile
                sys.path.insert(0, os.path.join(ROOT, "profiles", "tools"))
                import soak_driver
                d = tempfile.mkdtemp(prefix="rtbench_soak_")
                soak_driver.run(2, 4096, d)  # warm-up
                n_t = 8
                el = soak_driver.run(n_t, 4096, d)
                n_files = len(os.listdir(d))
                for f in os.listdir(d):
                    os.remove(os.path.join(d, f))
                os.rmdir(d)
                soak = {"frames_per_s": 2 * n_t / el, "frames": 2 * n_t, "seconds": el, "image": [4096, 4096], "spp": 64,
                        "mrays_per_s": 2 * n_t * 4096 * 4096 * 64 / el / 1e6, "files_written": n_files,
                        "includes": "per-timestep host BVH build (8 x 18,432 triangles, overlapped), upload, render, "
                                    "read-back, 50 MB P6 file per frame"}
            except Exception as e:
                soak = {"error": str(e)}
        # fp32 FAST PATH (RT_FLAG_FP32), reported separately with its measured tolerance against the exact
        # f64 frame of the same samples -- never part of `value`
        fp32 = None
        philox = None
        if not args.lean:
            try:
                frame(0)
                exact = ctx.download(rgb=False, hit_ids=want_ids)
                f32_ms = []
                for _ in range(3):
                    frame(0, fp32=True)
                    ctx.sync()
                    f32_ms.append(ctx.stats()["render_kernel_ms"])
                fast = ctx.download(rgb=False, hit_ids=want_ids)
                d = np.abs(fast["intensity"] - exact["intensity"])
                fp32 = {"kernel_ms": float(np.mean(f32_ms[1:])),
                        "kernel_only_mrays_s": rays_per_frame / (float(np.mean(f32_ms[1:])) * 1e-3) / 1e6,
                        "vs_f64_exact": {"max_abs_intensity_diff": float(d.max()), "mean_abs_intensity_diff": float(d.mean()),
                                         "hit_id_mismatch_fraction":
                                             float((fast["hit_prim"] != exact["hit_prim"]).mean()) if want_ids else None}}
                del exact, fast, d
            except Exception as e:  # the fast path must never take the bench line down
                fp32 = {"error": str(e)}
            # counter-based fast sampler (RT_SAMPLER_PHILOX): other samples than the reference, reported separately
            try:
                t_ph = []
                for _ in range(3):
                    ctx.sync()
                    t1 = time.perf_counter()
                    frame(0, philox=True)
                    ctx.sync()
                    t_ph.append(time.perf_counter() - t1)
                philox = {"ms_per_frame": float(np.mean(t_ph[1:])) * 1e3,
                          "mrays_per_s": rays_per_frame / float(np.mean(t_ph[1:])) / 1e6,
                          "note": "sampler kernel + traversal, no look-ahead needed; not the reference's samples"}
            except Exception as e:
                philox = {"error": str(e)}
        cpu = None
        if args.cpu_rays > 0 and world == 1:  # the CPU baseline is timed on rank 0 at N = 1 only
            frame(0)
            got = ctx.download(rgb=False, hit_ids=want_ids)
            cpu = cpu_baseline(coords, colors, cam, H, W, spp, got if want_ids else None, host, args.cpu_rays)
            # the oracle is an allocation-free C restatement; the reference itself (Eigen dynamic temporaries,
            # ~15 heap allocations per leaf visit) cannot be built on the GPU box.  SURVEY 6 [probed] timed it in
            # the survey container: 0.2-0.4 Mrays/s on one core, i.e. roughly a tenth of this port.
            cpu["reference_probed_mrays_s"] = [0.2, 0.4]
            cpu["reference_vs_port_note"] = ("kind=port: plain-C restatement of the reference's algorithm; the compiled "
                                             "reference measured 0.2-0.4 Mrays/s/core in the survey container "
                                             "(SURVEY.md 6, [probed]) -- about 10x slower than this port")
        total_rays = rays_per_frame * args.steps * world
        value = total_rays / elapsed / 1e6
        out = {
            "metric": "Mrays/s (primary+AA samples)",
            "value": value,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": label, "image": [H, W], "spp": spp, "triangles": int(host.info.n_tris),
                       "blas": int(host.info.n_meshes), "bvh_nodes": int(host.info.n_blas_nodes),
                       "sampler": "MT19937 reference stream (GPU GF(2) jump-ahead + jitter kernels, device-resident state)",
                       "hit_ids_written": want_ids, "frames_per_step_per_gpu": 1,
                       "cuts": ("off (RT_NO_CUT)" if os.environ.get("RT_NO_CUT") else
                                "made per frame: the whole hierarchy for frames >= %g rays, one level at the render tile from %g rays"
                                % (float(os.environ.get("RT_CUT_MIN_RAYS", 12e6)), float(os.environ.get("RT_CUT1_MIN_RAYS", 1e6)))),
                       "host_threads_per_rank": int(ra.lib.rt_host_threads()),
                       "parallelism": f"frame-sharded x{world}, no collective"},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "aux": {"jitter_kernel_ms": jms, "host_jump_ms_per_frame": hjm, "host_bvh_build_s": build_s,
                    "step_minus_traversal_kernel_ms": elapsed / args.steps * 1e3 - kms,
                    "upload_s": upload_s, "stack_need": int(host.info.stack_need),
                    "download_frame_ms": dl_frame_ms, "download_hit_ids_ms": dl_ids_ms,
                    "jitter_kernel_ms_timed_step": jitter_ms_last,
                    "end_to_end": end_to_end, "driver_4096x64": soak,
                    "multiframe": multiframe, "fp32_fast_path": fp32, "philox_sampler": philox, "single_frame_banded": banded},
        }
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out))
    ctx.close()


if __name__ == "__main__":
    main()
