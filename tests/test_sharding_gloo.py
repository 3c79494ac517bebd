"""N>1 path on CPU: two gloo ranks ask the product's frame scheduler (rt_shard_frames, the function
rt_render_scene itself uses) for their shards; together they must cover every frame exactly once, in the
reference's (timestep outer, camera inner) numbering, each at the right position of the never-reseeded RNG
stream -- and the generator state each rank would start from must equal sequential generation."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, T, C, frame_words, out):
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import orc
    import raytracer_amd as ra
    frames, pos = ra.shard_frames(T, C, rank, world, stream_base=0, frame_words=frame_words)
    # every frame's start state from the jump-ahead == sequential generation (checked on the first two)
    ok = 1
    for f, p in list(zip(frames, pos))[:2]:
        st = ra.mt19937_state_at(int(p))
        g = orc.Rng()
        g.set_state(st)
        h = orc.Rng()
        h.discard(int(p))
        ok &= int(np.array_equal(g.u32(64), h.u32(64)))
    mine = torch.full((T * C,), -1, dtype=torch.int64)
    mine[: len(frames)] = torch.from_numpy(frames.astype(np.int64))
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    flag = torch.tensor([ok])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        allf = torch.cat(gathered)
        allf = allf[allf >= 0]
        out.put((sorted(allf.tolist()), int(flag.item()), [int(x) for x in pos]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_cover_all_frames_once():
    world, T, C = 2, 5, 3
    frame_words = 18 * 30 * 2 * 4
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, T, C, frame_words, out)) for r in range(world)]
    for p in procs:
        p.start()
    frames, ok, pos0 = out.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert frames == list(range(T * C))
    assert ok == 1
    # T >= world: whole timesteps are dealt out (t % world == rank), so rank 0 holds t = 0, 2, 4 with all cameras
    assert pos0 == [(t * C + c) * frame_words for t in range(0, T, world) for c in range(C)]


def test_schedule_edge_cases():
    sys.path.insert(0, os.path.dirname(HERE))
    import raytracer_amd as ra
    f, p = ra.shard_frames(0, 2, 0, 1)
    assert len(f) == 0
    f, p = ra.shard_frames(3, 1, 3, 4, stream_base=100, frame_words=7)
    assert len(f) == 0
    f, p = ra.shard_frames(3, 2, 1, 4, stream_base=100, frame_words=7)  # fewer timesteps than ranks: by frame
    assert f.tolist() == [1, 5] and p.tolist() == [107, 135]
    f, p = ra.shard_frames(5, 2, 1, 4, stream_base=100, frame_words=7)  # enough timesteps: by timestep
    assert f.tolist() == [2, 3] and p.tolist() == [114, 121]
    for T, C, world in [(5, 3, 2), (8, 2, 8), (3, 2, 4), (100, 1, 8), (7, 2, 3)]:
        got = sorted(int(x) for r in range(world) for x in ra.shard_frames(T, C, r, world)[0])
        assert got == list(range(T * C))  # every frame exactly once
        if T >= world:
            for r in range(world):
                assert all((int(x) // C) % world == r for x in ra.shard_frames(T, C, r, world)[0])


# ----------------------------------------------------------------------------------------------
# One big frame split into row bands: schedule, stream positions and the gather.
# ----------------------------------------------------------------------------------------------
def _band_worker(rank, world, port, H, W, spp, frame_base, out):
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import orc
    import raytracer_amd as ra
    from raytracer_amd import meshgen
    c, k = meshgen.plate(24)
    cam = meshgen.camera_arrays(W, H)
    r0, r1, pos = ra.shard_rows(H, W, spp, rank, world, frame_base)
    # the band is rendered by the CPU oracle here (no GPU in this test): what is under test is the
    # product's schedule -- rows and stream position -- and the gather that assembles the frame
    rng = orc.Rng()
    rng.set_state(ra.mt19937_state_at(pos))
    band = orc.Scene([c], [k]).render(*cam, H, W, spp, rng=rng, row0=r0, row1=r1, want_hits=False)["intensity"]
    full = torch.zeros(H, W, dtype=torch.float64)
    full[r0:r1] = torch.from_numpy(band)
    rows = torch.zeros(H, dtype=torch.int64)
    rows[r0:r1] = 1
    dist.reduce(full, 0, op=dist.ReduceOp.SUM)  # disjoint bands: the sum is the gather
    dist.reduce(rows, 0, op=dist.ReduceOp.SUM)
    if rank == 0:
        g = orc.Rng()
        g.discard(frame_base)
        want = orc.Scene([c], [k]).render(*cam, H, W, spp, rng=g, want_hits=False)["intensity"]
        out.put((bool(np.array_equal(full.numpy().view(np.uint64), want.view(np.uint64))), rows.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_render_one_frame_in_row_bands():
    world, H, W, spp = 2, 20, 24, 3
    frame_base = 4 * 777
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_band_worker, args=(r, world, port, H, W, spp, frame_base, out)) for r in range(world)]
    for p in procs:
        p.start()
    same, rows = out.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert rows == [1] * H  # every row rendered exactly once
    assert same             # assembled frame == the single-rank frame, bit for bit


def test_row_band_schedule_properties():
    sys.path.insert(0, os.path.dirname(HERE))
    import raytracer_amd as ra
    for H, world in [(4096, 8), (225, 8), (7, 8), (64, 8), (1, 1), (100, 3), (2160, 7)]:
        W, spp, base = 13, 5, 1000
        prev = 0
        for r in range(world):
            r0, r1, pos = ra.shard_rows(H, W, spp, r, world, base)
            assert r0 == prev and r1 >= r0, (H, world, r)
            assert pos == base + r0 * W * spp * 4
            if H >= 8 * world and r > 0:
                assert r0 % 8 == 0
            prev = r1
        assert prev == H
    sizes = [ra.shard_rows(4096, 8, 1, r, 8)[1] - ra.shard_rows(4096, 8, 1, r, 8)[0] for r in range(8)]
    assert sizes == [512] * 8
    import pytest
    with pytest.raises(ra.RtError):
        ra.shard_rows(10, 10, 1, 3, 3)
