"""Parity soak on the GPU box: random scenes (1-8 meshes: plates, triangle soups, spheres, overlapping boxes so that
the TLAS slot quirk is often active), random cameras (inside and outside the scene, any direction), random image
shapes, row bands and sample counts on both sides of every instantiation boundary; each frame rendered through the
C ABI with per-sample outputs and compared with the oracle BIT FOR BIT (hit ids, t, intensity, rgb), plus, for a
third of the cases, the hit ids against the treeless argmin over all triangles.  The oracle is the checker here
(test infrastructure); nothing of it is timed.  Frames this small are walked from the root by default:
RT_CUT_MIN_RAYS=0 in the environment makes every frame take the cut walk (the path of the large frames),
RT_CUT1_MIN_RAYS=0 RT_CUT1_MAX_TRIS_PER_BLOCK=1e9 the one-level cuts of mid-sized frames (cut_single_kernel).
Usage: [RT_CUT_MIN_RAYS=0] python profiles/tools/soak_parity.py [cases] [first_seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc  # noqa: E402
import raytracer_amd as ra  # noqa: E402
from raytracer_amd import meshgen  # noqa: E402

SPPS = (1, 2, 3, 4, 5, 7, 8, 15, 16, 17, 31, 32, 33, 48, 49, 50, 56, 63, 64, 65, 100, 112, 127, 128, 130)


def bits(a):
    return np.ascontiguousarray(a, np.float64).view(np.uint64)


def random_scene(rs):
    coords, colors = [], []
    for m in range(int(rs.integers(1, 9))):
        kind = rs.integers(0, 4)
        x0, y0, z0 = rs.uniform(-5, 3), rs.uniform(-4, 2), rs.uniform(-9, -3)
        if kind == 0:
            c, k = meshgen.plate(int(rs.integers(2, 40)), side=float(rs.uniform(0.5, 6)), x0=x0, y0=y0, z0=z0,
                                 amp=float(rs.uniform(0, 0.4)))
        elif kind == 1:
            c, k = meshgen.random_soup(int(rs.integers(1, 3000)), seed=int(rs.integers(1 << 30)), extent=float(rs.uniform(0.5, 6)),
                                       size=float(rs.uniform(0.05, 1.5)), z_range=(z0 - 1.0, z0 + 1.0))
            c[0, :, :, 0] += x0
            c[0, :, :, 1] += y0
        elif kind == 2:
            c, k = meshgen.uv_sphere(int(rs.integers(3, 30)), int(rs.integers(3, 40)), radius=float(rs.uniform(0.3, 2.5)),
                                     center=(x0 + 2, y0 + 2, z0), color=float(rs.uniform(0.1, 0.9)))
        else:  # a clump of coincident / duplicate triangles: ties, extent-0 fallbacks, big leaves
            c, k = meshgen.random_soup(int(rs.integers(2, 40)), seed=int(rs.integers(1 << 30)), extent=1.0, size=0.8,
                                       z_range=(z0, z0))
            c[0, 1:] = c[0, :1] if rs.random() < 0.5 else c[0, 1:]
            c[0, :, :, 0] += x0
            c[0, :, :, 1] += y0
        coords.append(np.ascontiguousarray(c[:1]))
        colors.append(np.ascontiguousarray(k[:1]))
    return coords, colors


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    ctx = ra.Context(0)
    rays = hits = quirk = treeless = culled = from_cuts = 0
    by_spp = {}
    t_start = time.time()
    for case in range(n_cases):
        rs = np.random.default_rng(seed0 + case)
        coords, colors = random_scene(rs)
        host = ra.HostScene(coords, colors, 0, n_threads=int(rs.choice((0, 1, 4, -1))))
        osc = orc.Scene(coords, colors)
        t = host.tlas()
        order = [j for i in range(len(t["left"])) if t["left"][i] == -1
                 for j in range(t["min_blas_idx"][i], t["min_blas_idx"][i] + t["blas_count"][i])]
        quirk += order != sorted(order)
        ctx.upload(host)
        W, H = int(rs.integers(8, 200)), int(rs.integers(8, 120))
        spp = int(rs.choice(SPPS))
        if rs.random() < 0.25:  # a camera inside the cloud of meshes, looking anywhere
            center = (rs.uniform(-4, 4), rs.uniform(-3, 3), rs.uniform(-8, -3))
            target = tuple(np.array(center) + rs.normal(size=3))
        else:
            center = (rs.uniform(-3, 3), rs.uniform(-2, 2), rs.uniform(-1, 3))
            target = (rs.uniform(-3, 3), rs.uniform(-3, 3), -6.0)
        cam = meshgen.camera_arrays(W, H, center=center, target=target, fov=float(rs.uniform(20, 120)))
        r0 = int(rs.integers(0, H // 2)) // 8 * 8 if rs.random() < 0.3 else 0
        r1 = H if rs.random() < 0.7 else int(rs.integers(r0 + 1, H + 1))
        pos = int(rs.integers(0, 1 << 40)) * 4
        ctx.rng_seek(pos + r0 * W * spp * 4)
        ctx.render(*cam, H, W, spp, rows=(r0, r1), hit_ids=True, hit_t=True)
        got = ctx.download()
        from_cuts += ctx.stats()["finest_cut_level"] >= 0
        rng = orc.Rng()
        rng.set_state(ra.mt19937_state_at(pos + r0 * W * spp * 4))
        want = osc.render(*cam, H, W, spp, rng=rng, row0=r0, row1=r1)
        mesh, elem = host.map_prims(got["hit_prim"])
        what = f"case {case} (seed {seed0 + case}): {len(coords)} meshes, {W}x{H} rows {r0}:{r1}, {spp} spp"
        assert np.array_equal(mesh, want["hit_mesh"]) and np.array_equal(elem, want["hit_elem"]), what
        assert np.array_equal(bits(got["hit_t"]), bits(want["hit_t"])), what
        assert np.array_equal(bits(got["intensity"]), bits(want["intensity"])), what
        assert np.array_equal(got["rgb"], want["rgb"]), what
        n = (r1 - r0) * W * spp
        rays += n
        hits += int((want["hit_mesh"] >= 0).sum())
        by_spp[spp] = by_spp.get(spp, 0) + 1
        if case % 3 == 0:
            # builder- and walk-independent: the closest of ALL triangles; a disagreement must be a triangle the
            # reference's own box tests cull for that ray (tests/test_treeless.py: explain_disagreements)
            import test_treeless as tt
            rng.set_state(ra.mt19937_state_at(pos + r0 * W * spp * 4))
            dirs = orc.ray_dirs(*cam, W, spp, rng.doubles(n * 2) - 0.5, row0=r0, row1=r1)
            tri, mesh_of, elem_of = tt.all_tris(coords, 0)
            bt, first, nmin = orc.brute_force(cam[0], dirs, tri)
            hit = first >= 0
            bm = np.where(hit, mesh_of[np.maximum(first, 0)], -1)
            be = np.where(hit, elem_of[np.maximum(first, 0)], -1)
            same = (bm == mesh) & (be == elem) & (bits(bt) == bits(got["hit_t"]))
            tied = (~same) & (nmin > 1) & (bits(bt) == bits(got["hit_t"]))
            bad = np.nonzero(~same & ~tied)[0]
            if bad.size:
                n_expl = tt.explain_disagreements(bad[:2000], cam[0], dirs, osc, mesh_of, elem_of, first)
                assert n_expl == min(bad.size, 2000), (what, int(bad.size), n_expl)
                culled += int(bad.size)
            treeless += 1
        if case % 20 == 19:
            print(f"{case + 1} cases, {rays / 1e6:.1f} M rays, {time.time() - t_start:.0f} s", flush=True)
    ctx.close()
    print(f"{n_cases} random frames identical to the oracle bit for bit (hit ids, t, intensity, rgb): {rays} rays, "
          f"hit rate {hits / max(rays, 1):.3f}, {from_cuts} frames walked from per-block cuts, TLAS slot quirk active in {quirk} scenes, treeless argmin checked on {treeless} frames "
          f"({culled} rays whose closest triangle a box of the reference's own walk culls, each one explained); "
          f"sample counts used: {sorted(by_spp)}")


if __name__ == "__main__":
    main()
