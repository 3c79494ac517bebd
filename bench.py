#!/usr/bin/env python3
"""bench.py -- Mrays/s of the render hot path on MI355X (contract: see the round prompt / DESIGN.md).

A "step" is one full frame of the workload: MT19937 jitter generation (GPU GF(2) jump-ahead kernels +
jitter kernel, the host only supplies cached jump polynomials), ray generation, TLAS/BLAS traversal, Moller-Trumbore, shading, in-order sample accumulation,
quantisation and per-sample hit-id output, with the flattened tree already resident in HBM (the host
BVH build and the upload happen before the timed region and are reported separately).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c4|plate1m|sphere|block]

N > 1 is launched by torch.distributed.run, one rank per GPU; frames are sharded over ranks
(frame f = step*N + rank, each at its own position of the reference's global RNG stream), with no
data-path collective: only the barrier and the MAX-over-ranks of the elapsed time.
Prints ONE JSON line on rank 0.  Besides the contract's fields it carries, under "aux" and never as part of
`value`: the multi-frame driver's frames/s (BASELINE's second metric), the fp32 fast path with its measured
tolerance, and for N > 1 one frame split into row bands over the ranks (strong scaling) with its gather.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
TDEV = "cuda"  # device of the tensors handed to torch.distributed (see RT_BENCH_BACKEND)


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


def bench_multiframe(ra, rank, world, local_rank, dist, steps_per_rank=16):
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

    run(min(T, 2 * world))  # warm-up: polynomials, allocations
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    run(T)
    elapsed = time.perf_counter() - t0
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
    return {"frames_per_s": frames / elapsed, "frames": frames, "seconds": elapsed, "image": [H, W], "spp": spp,
            "meshes": 8, "triangles": int(counts.sum()), "cameras": 2, "timesteps": T,
            "mrays_per_s": frames * H * W * spp / elapsed / 1e6, "files_written_by_rank0": n_files,
            "includes": "per-timestep host BVH build (overlapped), upload, render, read-back, P6 write"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("RT_BENCH_WORKLOAD", "c4"))
    ap.add_argument("--cpu-rays", type=float, default=40e6, help="rays in the CPU-baseline sample, about 10 s of one core (0 = skip)")
    ap.add_argument("--no-hit-ids", action="store_true")
    ap.add_argument("--no-multiframe", action="store_true", help="skip the frames/s (multi-frame driver) measurement")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
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
    kernel_ms = []
    jitter_ms = []
    t0 = time.perf_counter()
    for s in range(args.steps):
        frame(args.warmup + s)
        # stats of the PREVIOUS launch would stall the pipeline; collect after the loop for the last one,
        # and per step only when the render has to be waited for anyway (rt_render syncs at entry)
        st = None
        if s == args.steps - 1:
            ctx.sync()
            st = ctx.stats()
        if st:
            kernel_ms.append(st["render_kernel_ms"])
            jitter_ms.append(st["jitter_kernel_ms"])
    barrier()
    elapsed = time.perf_counter() - t0
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
    # (RCCL gather of the f64 intensity bands).  Reported in aux, never part of `value`.
    banded = None
    if world > 1:
        try:
            import torch
            r0, r1, pos = ra.shard_rows(H, W, spp, rank, world, 0)

            def band_step():
                if r1 > r0:
                    ctx.rng_seek(pos)
                    ctx.render(*cam, H, W, spp, rows=(r0, r1), hit_ids=want_ids)

            band_step()
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                band_step()
            barrier()
            t_render = (time.perf_counter() - t0) / args.steps
            t0 = time.perf_counter()
            band = ctx.download(rgb=False, hit_ids=False)["intensity"] if r1 > r0 else np.empty((0, W))
            sizes = [ra.shard_rows(H, W, spp, r, world, 0) for r in range(world)]
            pad = max(b - a for a, b, _ in sizes)
            mine = torch.zeros(pad, W, dtype=torch.float64, device=TDEV)
            mine[: r1 - r0] = torch.from_numpy(band).to(TDEV)
            parts = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
            dist.gather(mine, parts, dst=0)
            torch.cuda.synchronize()
            t_gather = time.perf_counter() - t0
            tt = torch.tensor([t_render, t_gather], dtype=torch.float64, device=TDEV)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            banded = {"ms_per_frame_render": float(tt[0]) * 1e3, "mrays_per_s": rays_per_frame / float(tt[0]) / 1e6,
                      "download_plus_gather_ms": float(tt[1]) * 1e3, "bands": [[a, b] for a, b, _ in sizes],
                      "scaling": "strong"}
        except Exception as e:
            banded = {"error": str(e)}

    # untimed extras on rank 0: per-launch kernel time over several launches, visit counters, CPU baseline
    out = None
    if rank == 0:
        ks = []
        for s in range(max(2, min(args.steps, 3))):
            frame(0)
            ctx.sync()
            ks.append(ctx.stats())
        kms = float(np.mean([k["render_kernel_ms"] for k in ks]))
        jms = float(np.mean([k["jitter_kernel_ms"] for k in ks]))
        hjm = float(np.mean([k["host_jump_ms"] for k in ks]))
        frame(0, counters=True)
        ctx.sync()
        cs = ctx.stats()
        bray = b_ray(cs["nodes_visited"], cs["tris_tested"], cs["rays_hit"], cs["rays"], spp)
        achieved = rays_per_frame * bray / (kms * 1e-3) / 1e9
        traffic = None
        valu = None
        pmc_path = os.path.join(ROOT, "profiles", f"pmc_{args.workload}.json")
        if os.path.exists(pmc_path):
            try:
                pmc = json.load(open(pmc_path))
                traffic = pmc.get("hbm_bytes_per_launch")
                # the bound that actually binds: non-fused fp64 VALU issue, 256 CU x 4 SIMD x
                # 16 lanes x 2.4 GHz = 39.3 T lane-ops/s; instruction count from the committed PMC pass
                vi = pmc.get(pmc.get("dominant_kernel", ""), {}).get("SQ_INSTS_VALU")
                if vi:
                    valu = {"lane_ops_per_launch": vi * 64, "peak_lane_ops_per_s": 39.3e12,
                            "frac_of_fp64_issue_peak": vi * 64 / (kms * 1e-3) / 39.3e12,
                            "source": "profiles/pmc_%s.json (SQ_INSTS_VALU x 64)" % args.workload}
            except Exception:
                traffic = None
        # host hand-over cost (PCIe), never part of `value`
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
        # fp32 FAST PATH (RT_FLAG_FP32), reported separately with its measured tolerance against the exact
        # f64 frame of the same samples -- never part of `value`
        fp32 = None
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
        philox = None
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
            if want_ids:
                cpu = cpu_baseline(coords, colors, cam, H, W, spp, got, host, args.cpu_rays)
            else:
                cpu = cpu_baseline(coords, colors, cam, H, W, spp, None, host, args.cpu_rays)
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
                       "parallelism": f"frame-sharded x{world}, no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "rtk::render_packet_kernel", "kernel_ms": kms, "bytes_per_ray": bray,
                         "rays_per_launch": rays_per_frame, "nodes_per_ray": cs["nodes_visited"] / cs["rays"],
                         "tris_per_ray": cs["tris_tested"] / cs["rays"], "hit_rate": cs["rays_hit"] / cs["rays"],
                         "kernel_only_mrays_s": rays_per_frame / (kms * 1e-3) / 1e6,
                         "wave_node_fetches_per_ray": cs["wave_node_fetches"] / cs["rays"],
                         "valu": valu},
            "cpu_baseline": cpu,
            "aux": {"jitter_kernel_ms": jms, "host_jump_ms_per_frame": hjm, "host_bvh_build_s": build_s,
                    "upload_s": upload_s, "stack_need": int(host.info.stack_need),
                    "download_frame_ms": dl_frame_ms, "download_hit_ids_ms": dl_ids_ms,
                    "kernel_ms_timed_step": kernel_ms[-1] if kernel_ms else None,
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
