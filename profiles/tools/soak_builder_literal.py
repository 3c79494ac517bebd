"""Builder soak on the CPU: random scenes -- soups with exact duplicates and unsplittable clumps, regular grids (rows of
equal centroid coordinates, centroids on bin borders; stored in row order, so the run binning takes them), thin sheets,
coordinates full of +0.0 / -0.0, 1-6 meshes -- built by the line-by-line statement of rtbvh.cpp in binary64
(oracle/derive_tree_fixtures.py), by the oracle's builder and by the product's (1 thread, 4 threads, one thread per
tree): TLAS, slot order, every BLAS node (boxes bit for bit), sorted order and message count must all be the literal
statement's.  The literal statement and the oracle are the checkers (test infrastructure).
Usage: python profiles/tools/soak_builder_literal.py [scenes] [first_seed]"""
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
import test_tree_fixtures as tf  # noqa: E402


def random_mesh(rs):
    kind = rs.integers(0, 5)
    if kind == 0:
        c, _ = meshgen.random_soup(int(rs.integers(5, 2500)), seed=int(rs.integers(1 << 30)), extent=float(rs.uniform(0.5, 8)),
                                   size=float(rs.uniform(0.05, 1.0)))
        n = c.shape[1]
        if n > 40 and rs.random() < 0.6:
            k = int(rs.integers(1, n // 4))
            c[0, n - k:] = c[0, :k]  # exact duplicates: cost ties
        if n > 20 and rs.random() < 0.4:
            c[0, 5:5 + int(rs.integers(5, 12))] = c[0, 0]  # a clump: extent 0 -> midpoint -> abort
    elif kind == 1:
        c, _ = meshgen.plate(int(rs.integers(3, 40)), side=float(rs.uniform(1, 6)), amp=float(rs.uniform(0, 0.3)))
    elif kind == 2:
        c, _ = meshgen.uv_sphere(int(rs.integers(4, 30)), int(rs.integers(4, 40)), radius=float(rs.uniform(0.3, 2.5)))
    elif kind == 3:
        c, _ = meshgen.random_soup(int(rs.integers(50, 1500)), seed=int(rs.integers(1 << 30)), extent=3.0, size=0.4)
        c[0, :, :, int(rs.integers(0, 3))] *= 10.0 ** -int(rs.integers(6, 12))  # nearly planar
    else:
        c, _ = meshgen.plate(int(rs.integers(8, 32)), side=4.0)  # flat regular grid, then zeros of both signs
        flat = c.reshape(-1)
        flat[rs.integers(0, flat.size, flat.size // 8)] = 0.0
        flat[rs.integers(0, flat.size, flat.size // 8)] = -0.0
    c = np.ascontiguousarray(c[:1])
    c[0] += rs.uniform(-6, 6, 3) * (rs.random() < 0.8)
    return c


def main():
    n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 9000
    lit_mod = tf.literal_statement()
    tris = nodes = msgs = quirk = 0
    t0 = time.time()
    for i in range(n_scenes):
        rs = np.random.default_rng(seed0 + i)
        meshes = [random_mesh(rs) for _ in range(int(rs.integers(1, 7)))]
        lit = lit_mod.build_scene_binary64([m[0] for m in meshes])
        colors = [np.full((1, c.shape[1], 3), 0.5) for c in meshes]

        def nodes_of(ns):
            return dict(boxes=[n["box"].six() for n in ns], count=[n["count"] for n in ns], left=[n["left"] for n in ns],
                        min_idx=[n["min_idx"] for n in ns])

        want_tlas = nodes_of(lit["tlas"])
        builders = [("oracle", lambda: orc.Scene(meshes, colors))]
        for nt in (1, 4, -1):
            builders.append((f"product n_threads={nt}", lambda nt=nt: ra.HostScene(meshes, colors, 0, n_threads=nt)))
        for who, make in builders:
            sc = make()
            what = f"scene {i} (seed {seed0 + i}), {who}"
            t = sc.tlas()
            tf.check_nodes(dict(boxes=t["boxes"], left=t["left"], count=t["blas_count"], min_idx=t["min_blas_idx"]), want_tlas,
                           f"{what}: TLAS")
            got_msgs = sc.midpoint_msgs if who == "oracle" else sc.info.midpoint_fallbacks
            assert got_msgs == lit["messages"], what
            for s, mesh in enumerate(lit["slots"]):
                got = sc.slot(s)
                # (check_nodes compares values; here also the bits: the sign of a zero extreme)
                assert np.array_equal(np.ascontiguousarray(got["boxes"], np.float64).view(np.uint64),
                                      np.array([n["box"].six() for n in lit["blases"][mesh]["nodes"]], np.float64).view(np.uint64)), \
                    f"{what}: box bits of mesh {mesh}"
                assert got["mesh"] == mesh, f"{what}: slot {s}"
                want = lit["blases"][mesh]
                tf.check_nodes(got, nodes_of(want["nodes"]), f"{what}: slot {s} = mesh {mesh}")
                assert list(got["indices"]) == want["indices"], f"{what}: sorted order of mesh {mesh}"
        tris += sum(m.shape[1] for m in meshes)
        nodes += sum(len(b["nodes"]) for b in lit["blases"])
        msgs += lit["messages"]
        quirk += lit["slots"] != sorted(lit["slots"])
        if i % 10 == 9:
            print(f"{i + 1} scenes, {tris} triangles, {time.time() - t0:.0f} s", flush=True)
    print(f"{n_scenes} random scenes: the oracle's builder and the product's (1 thread, 4 threads, one thread per tree) "
          f"build the literal statement's trees node for node -- {tris} triangles, {nodes} BLAS nodes, {msgs} midpoint "
          f"messages, slot order differs from mesh order in {quirk} scenes")


if __name__ == "__main__":
    main()
