"""BASELINE.json configs[0], [1], [2] and [4] at the reference's OWN sizes -- 400 x 225 images (rttests.py:31-33), the
loop order, stream and file names of render_scene (cpp/rtmain.cpp:49-64) -- on the GPU through the C ABI, every frame
against the oracle.  (configs[3] at its own size: tests/test_gpu_configs.py.)

  configs[0]  simple_block.e, 1 camera, 1 spp, all 9 timesteps through rt_render_scene: 9 P6 files
  configs[1]  the VTK sphere, 50 spp: every sample's hit id and t, every pixel's intensity bits, the P6 payload
  configs[2]  a 100-timestep displacement sequence sharded over world = 8: the eight ranks' schedules run one after the
              other on the one GPU of this box (one process per rank, as under torchrun), 100 files in one directory
  configs[4]  2 cameras x 8 meshes x 50 spp, tree rebuilt per timestep, TLAS slot quirk active, 16 timesteps through
              the driver: 32 files; and per-sample hit ids of the first and last timestep
The oracle renders the frames of a job on all host cores: each frame starts at its own position of the one continuous
MT19937 stream (frame f of a job at base + f * H*W*spp*4 words).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import orc
import raytracer_amd as ra
import scenes
from raytracer_amd import _lib as L

pytestmark = pytest.mark.gpu
H, W = scenes.H, scenes.W
HDR = f"P6\n{W} {H}\n255\n".encode()


def bits(a):
    return np.ascontiguousarray(a, np.float64).view(np.uint64)


def oracle_job(coords, colors, cams, spp, T, base, want_hits=False):
    """The frames of render_scene(T timesteps x cameras) from stream position `base` -> {(t, cam): oracle output}."""
    words = H * W * spp * 4
    built = {}

    def one(f):
        t, c = divmod(f, len(cams))
        if t not in built:  # (racing threads may build a timestep twice: harmless)
            built[t] = orc.Scene(coords, colors, t)
        rng = orc.Rng()
        rng.set_state(ra.mt19937_state_at(base + f * words))
        return (t, c), built[t].render(*cams[c], H, W, spp, rng=rng, want_hits=want_hits)

    with ThreadPoolExecutor(orc.n_threads()) as ex:
        return dict(ex.map(one, range(T * len(cams))))


def check_files(out, want):
    assert sorted(os.listdir(out)) == sorted(f"rtimage_{t}_cam{c}.ppm" for t, c in want)
    for (t, c), w in want.items():
        assert (out / f"rtimage_{t}_cam{c}.ppm").read_bytes() == HDR + w["rgb"].tobytes(), (t, c)


def test_configs0_block_9_timesteps_through_the_driver(tmp_path):
    coords, colors, cams, spp = scenes.block()
    base = int(L.lib.rt_driver_stream_position(0))
    ra.render_scene(H, W, scenes.as_scene(coords, colors, cams), spp, tmp_path)
    want = oracle_job(coords, colors, cams, spp, 9, base)
    check_files(tmp_path, want)
    assert len({w["rgb"].tobytes() for w in want.values()}) == 9  # the block really deforms from frame to frame


def test_configs1_sphere_50spp_every_sample():
    coords, colors, cams, spp = scenes.sphere()
    host = ra.HostScene(coords, colors, 0)
    ctx = ra.Context(0)
    try:
        ctx.upload(host)
        ctx.rng_seed(5489)
        ctx.render(*cams[0], H, W, spp, hit_ids=True, hit_t=True)
        g = ctx.download(pinned=False)
    finally:
        ctx.close()
    w = orc.Scene(coords, colors, 0).render(*cams[0], H, W, spp, rng=orc.Rng())
    mesh, elem = host.map_prims(g["hit_prim"])
    assert g["hit_prim"].size == 4_500_000
    assert np.array_equal(mesh, w["hit_mesh"]) and np.array_equal(elem, w["hit_elem"])
    assert np.array_equal(bits(g["hit_t"]), bits(w["hit_t"]))
    assert np.array_equal(bits(g["intensity"]), bits(w["intensity"]))
    assert np.array_equal(g["rgb"], w["rgb"])
    assert 0.1 < (w["hit_elem"] >= 0).mean() < 0.9


_RANK_SCRIPT = r'''
import os, sys
root, out, rank, world, T = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5])
sys.path[:0] = [root, os.path.join(root, "tests")]
os.environ["RT_SHARD_FRAMES"] = "1"; os.environ["RANK"] = rank; os.environ["WORLD_SIZE"] = world
os.environ["LOCAL_RANK"] = "0"  # one GPU on this box: the ranks take turns on it
import raytracer_amd as ra, scenes
coords, colors, cams, spp = scenes.sequence(T)
ra.render_scene(scenes.H, scenes.W, scenes.as_scene(coords, colors, cams), spp, out)
'''


def test_configs2_100_timesteps_sharded_over_8_ranks(tmp_path):
    T, world = 100, 8
    coords, colors, cams, spp = scenes.sequence(T)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT)
    out = tmp_path / "frames"
    out.mkdir()
    for rank in range(world):  # one after the other: at most one extra process on the GPU at a time
        subprocess.check_call([sys.executable, str(script), root, str(out), str(rank), str(world), str(T)], timeout=300,
                              stdout=subprocess.DEVNULL)
        mine, _ = ra.shard_frames(T, 1, rank, world)
        assert len(os.listdir(out)) == sum(len(ra.shard_frames(T, 1, r, world)[0]) for r in range(rank + 1)), rank
        assert all((out / f"rtimage_{int(f)}_cam0.ppm").exists() for f in mine)
    want = oracle_job(coords, colors, cams, spp, T, 0)  # every rank is a fresh process: its stream starts at 0
    check_files(out, want)
    assert len({w["rgb"].tobytes() for w in want.values()}) == T


def test_configs4_stereo_8_meshes_50spp_16_timesteps_through_the_driver(tmp_path):
    T = 16
    coords, colors, cams, spp = scenes.stereo(T)
    for t in range(T):  # the slot quirk is active at every timestep
        tl = ra.HostScene(coords, colors, t).tlas()
        order = [j for i in range(len(tl["left"])) if tl["left"][i] == -1
                 for j in range(tl["min_blas_idx"][i], tl["min_blas_idx"][i] + tl["blas_count"][i])]
        assert order != sorted(order), t
    base = int(L.lib.rt_driver_stream_position(0))
    ra.render_scene(H, W, scenes.as_scene(coords, colors, cams), spp, tmp_path)
    assert int(L.lib.rt_driver_stream_position(0)) == base + T * 2 * H * W * spp * 4
    want = oracle_job(coords, colors, cams, spp, T, base, want_hits=True)
    check_files(tmp_path, want)
    seen = set()
    for w in want.values():
        seen |= set(np.unique(w["hit_mesh"]).tolist())
    assert len(seen - {-1}) >= 7
    # per-sample ids and t of the first and the last timestep, both cameras, through rt_render at the same positions
    ctx = ra.Context(0)
    try:
        for t in (0, T - 1):
            host = ra.HostScene(coords, colors, t)
            ctx.upload(host)
            for c in range(2):
                ctx.rng_seed(5489)
                ctx.rng_seek(base + (t * 2 + c) * H * W * spp * 4)
                ctx.render(*cams[c], H, W, spp, hit_ids=True, hit_t=True)
                g = ctx.download(pinned=False)
                mesh, elem = host.map_prims(g["hit_prim"])
                w = want[(t, c)]
                assert np.array_equal(mesh, w["hit_mesh"]) and np.array_equal(elem, w["hit_elem"]), (t, c)
                assert np.array_equal(bits(g["hit_t"]), bits(w["hit_t"])), (t, c)
                assert np.array_equal(bits(g["intensity"]), bits(w["intensity"])), (t, c)
    finally:
        ctx.close()
