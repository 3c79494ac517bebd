"""Parity soak of the whole-job driver on the GPU box: random jobs -- 1-8 deforming meshes (plates that sway, spheres
that drift, soups that breathe; meshes whose trees gain and lose nodes from timestep to timestep), 1-3 cameras, 2-9
timesteps, image sizes and sample counts on both sides of the cut thresholds -- through render_scene (the builds of
several timesteps in flight on the pool, staged uploads, pipelined frames, grouped sampler launches, P6 writer
thread), every file compared byte for byte with the oracle rendering the same frames over ONE continuous stream.
Every third job is sharded: two "ranks" render their schedules one after the other into the same directory.
The oracle is the checker here (test infrastructure); nothing of it is timed.
Usage: python profiles/tools/soak_driver_parity.py [jobs] [first_seed]"""
import os
import shutil
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc  # noqa: E402
import raytracer_amd as ra  # noqa: E402
from raytracer_amd import _lib as L, meshgen  # noqa: E402


def random_job(rs):
    T = int(rs.integers(2, 10))
    coords, colors = [], []
    for m in range(int(rs.integers(1, 9))):
        kind = rs.integers(0, 3)
        x0, y0, z0 = rs.uniform(-5, 2), rs.uniform(-4, 1), rs.uniform(-8, -3)
        if kind == 0:
            c, k = meshgen.plate(int(rs.integers(2, 48)), side=float(rs.uniform(1, 6)), x0=x0, y0=y0, z0=z0, T=T,
                                 sway=float(rs.uniform(0, 0.05)))
        elif kind == 1:
            c, k = meshgen.uv_sphere(int(rs.integers(3, 24)), int(rs.integers(3, 30)), radius=float(rs.uniform(0.3, 2.0)),
                                     center=(x0 + 2, y0 + 2, z0), color=float(rs.uniform(0.1, 0.9)), T=T)
            c = c + rs.uniform(-0.3, 0.3, 3)[None, None, None, :] * np.linspace(0.0, 1.0, T)[:, None, None, None]
        else:
            c, k = meshgen.random_soup(int(rs.integers(1, 2500)), seed=int(rs.integers(1 << 30)), extent=float(rs.uniform(0.5, 5)),
                                       size=float(rs.uniform(0.05, 1.2)), z_range=(z0 - 1.0, z0 + 1.0), T=T)
            c = c * (1.0 + 0.02 * np.arange(T))[:, None, None, None]
            c[..., 0] += x0
            c[..., 1] += y0
        coords.append(np.ascontiguousarray(c))
        colors.append(np.ascontiguousarray(k))
    W, H = int(rs.integers(16, 260)), int(rs.integers(16, 160))
    spp = int(rs.choice((1, 2, 5, 16, 31, 50, 64, 65, 100)))
    if W * H * spp > 3_000_000:  # keep the oracle's share of the run short
        spp = max(1, 3_000_000 // (W * H))
    cams = []
    for _ in range(int(rs.integers(1, 4))):
        cams.append(meshgen.camera_arrays(W, H, center=(rs.uniform(-2, 2), rs.uniform(-2, 2), rs.uniform(-1, 2)),
                                          target=(rs.uniform(-2, 2), rs.uniform(-2, 2), -6.0), fov=float(rs.uniform(30, 110))))
    return coords, colors, cams, T, H, W, spp


def as_scene(coords, colors, cams, T):
    sc = ra.Scene()
    for c, k in zip(coords, colors):
        sc.add_mesh(c, None, None, k, T)
    for cam in cams:
        sc.add_camera(*cam)
    return sc


def main():
    n_jobs = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 7000
    frames = rays = sharded = with_cuts = 0
    t_start = time.time()
    for job in range(n_jobs):
        rs = np.random.default_rng(seed0 + job)
        coords, colors, cams, T, H, W, spp = random_job(rs)
        out = tempfile.mkdtemp(prefix="rtsoak_")
        words = H * W * spp * 4
        n_frames = T * len(cams)
        frame_base = {}  # frame -> stream position the job it was rendered by started from
        saved, devnull = os.dup(1), os.open(os.devnull, os.O_WRONLY)
        sys.stdout.flush()
        os.dup2(devnull, 1)
        try:
            if job % 3 == 2:
                # two ranks' schedules, one call after the other in this process: each call renders its own frames at
                # their positions of the stream IT started from and leaves the stream a whole job further on
                sharded += 1
                for rank in range(2):
                    base = int(L.lib.rt_driver_stream_position(0))
                    mine, _ = ra.shard_frames(T, len(cams), rank, 2, base, words)
                    for f in mine:
                        frame_base[int(f)] = base
                    os.environ.update(RT_SHARD_FRAMES="1", RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0")
                    ra.render_scene(H, W, as_scene(coords, colors, cams, T), spp, out)
                    assert int(L.lib.rt_driver_stream_position(0)) == base + n_frames * words, "stream position after the job"
            else:
                base = int(L.lib.rt_driver_stream_position(0))
                frame_base = {f: base for f in range(n_frames)}
                ra.render_scene(H, W, as_scene(coords, colors, cams, T), spp, out)
                assert int(L.lib.rt_driver_stream_position(0)) == base + n_frames * words, "stream position after the job"
        finally:
            for k in ("RT_SHARD_FRAMES", "RANK", "WORLD_SIZE", "LOCAL_RANK"):
                os.environ.pop(k, None)
            os.dup2(saved, 1)
            os.close(saved)
            os.close(devnull)
        assert sorted(frame_base) == list(range(n_frames)), "the two schedules cover every frame once"
        built = {}

        def one(f):
            t, c = divmod(f, len(cams))
            if t not in built:
                built[t] = orc.Scene(coords, colors, t)
            rng = orc.Rng()
            rng.set_state(ra.mt19937_state_at(frame_base[f] + f * words))
            return (t, c), built[t].render(*cams[c], H, W, spp, rng=rng, want_hits=False)["rgb"].tobytes()

        with ThreadPoolExecutor(orc.n_threads()) as ex:
            want = dict(ex.map(one, range(n_frames)))
        what = f"job {job} (seed {seed0 + job}): {len(coords)} meshes, {len(cams)} cameras, {T} timesteps, {W}x{H}x{spp}"
        assert sorted(os.listdir(out)) == sorted(f"rtimage_{t}_cam{c}.ppm" for t, c in want), what
        hdr = f"P6\n{W} {H}\n255\n".encode()
        for (t, c), rgb in want.items():
            with open(os.path.join(out, f"rtimage_{t}_cam{c}.ppm"), "rb") as f:
                assert f.read() == hdr + rgb, (what, t, c)
        shutil.rmtree(out)
        frames += len(want)
        rays += len(want) * H * W * spp
        with_cuts += H * W * spp >= 1_000_000
        if job % 10 == 9:
            print(f"{job + 1} jobs, {frames} frames, {rays / 1e6:.0f} M rays, {time.time() - t_start:.0f} s", flush=True)
    print(f"{n_jobs} random jobs through render_scene, {frames} P6 files byte-identical to the oracle's continuous stream "
          f"({rays} rays; {sharded} jobs as two ranks' schedules; {with_cuts} jobs of frames >= 1 M rays)")


if __name__ == "__main__":
    main()
