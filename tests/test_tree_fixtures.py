"""The hand-derivable tree fixtures (tests/golden/tree_fixtures.json, derivation beside it, generator
oracle/derive_tree_fixtures.py): a third, line-by-line statement of the reference's builder
(cpp/rtbvh.cpp:132-476, 544-561) on inputs small enough to check with pencil and paper, evaluated in binary64 AND in
exact rational arithmetic with every decision required to agree.  Both builders -- the oracle's (oracle/rt_oracle.c)
and the product's (raytracer_amd/csrc/rt_bvh.cpp) -- must reproduce the expected TLAS nodes, the slot order of
tlas.blases (quirk 1), every BLAS node, the sorted index order (including the rotation an ABORTED partition leaves
behind) and the number of "SAH splitting failed" messages; the oracle's walk and the GPU's must report the hand-derived
hit of every probe ray (which meshes a TLAS leaf really guards; first of bit-equal t in stored order).
"""
import json
import os

import numpy as np
import pytest

import orc
import raytracer_amd as ra


def load():
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tree_fixtures.json")) as f:
        return json.load(f)


FIXTURES = load()


def arrays(fx):
    coords = [np.ascontiguousarray(np.array(m, np.float64).reshape(1, -1, 3, 3)) for m in fx["meshes"]]
    colors = [np.full((1, c.shape[1], 3), 0.25 + 0.05 * i) for i, c in enumerate(coords)]
    return coords, colors


def check_nodes(got, want, what):
    assert np.array_equal(np.asarray(got["boxes"], np.float64), np.array(want["boxes"], np.float64)), what
    assert list(got["left"]) == want["left"], what
    assert [int(c) for c in got["count"]] == want["count"], what
    assert list(got["min_idx"]) == want["min_idx"], what


@pytest.mark.parametrize("fx", FIXTURES, ids=[f["name"] for f in FIXTURES])
@pytest.mark.parametrize("who", ["oracle", "product", "product_4_threads"])
def test_builder_reproduces_the_hand_derived_tree(fx, who):
    coords, colors = arrays(fx)
    if who == "oracle":
        sc = orc.Scene(coords, colors)
        n_slots, msgs = sc.n_slots, sc.midpoint_msgs
    else:
        sc = ra.HostScene(coords, colors, 0, n_threads=4 if who.endswith("threads") else 1)
        n_slots, msgs = sc.info.n_meshes, sc.info.midpoint_fallbacks
    t = sc.tlas()
    check_nodes(dict(boxes=t["boxes"], left=t["left"], count=t["blas_count"], min_idx=t["min_blas_idx"]), fx["tlas"],
                "TLAS")
    assert n_slots == len(fx["slot_to_mesh"])
    for s, mesh in enumerate(fx["slot_to_mesh"]):
        got = sc.slot(s)
        assert got["mesh"] == mesh, f"slot {s}"
        want = fx["blases"][mesh]
        check_nodes(got, want, f"slot {s} = mesh {mesh}")
        assert list(got["indices"]) == want["indices"], f"sorted order of mesh {mesh}"
    assert msgs == fx["midpoint_messages"]


def probe_arrays(fx):
    o = np.array([p["origin"] for p in fx["probes"]], np.float64)
    d = np.array([p["dir"] for p in fx["probes"]], np.float64)
    mesh = np.array([p["expect"]["mesh"] for p in fx["probes"]], np.int32)
    elem = np.array([p["expect"]["elem"] for p in fx["probes"]], np.int32)
    t = np.array([np.inf if p["expect"]["t"] is None else p["expect"]["t"] for p in fx["probes"]])
    return o, d, mesh, elem, t


@pytest.mark.parametrize("fx", FIXTURES, ids=[f["name"] for f in FIXTURES])
def test_oracle_walk_reports_the_hand_derived_hits(fx):
    coords, colors = arrays(fx)
    o, d, mesh, elem, t = probe_arrays(fx)
    got = orc.Scene(coords, colors).trace(o, d)
    assert np.array_equal(got["mesh"], mesh) and np.array_equal(got["elem"], elem)
    hit = mesh >= 0
    assert np.all(np.isinf(got["t"][~hit])) and np.allclose(got["t"][hit], t[hit], rtol=0, atol=1e-12)
    assert hit.any() and (~hit).any() or fx["name"] == "blas11"


def test_fixtures_take_the_branches_they_were_made_for():
    by = {f["name"]: f for f in FIXTURES}
    for name in ("tlas5", "tlas8"):  # quirk 1: the leaves do not appear in range order
        t = by[name]["tlas"]
        order = [j for i in range(len(t["left"])) if t["left"][i] == -1
                 for j in range(t["min_idx"][i], t["min_idx"][i] + t["count"][i])]
        assert order != sorted(order), name
    b = by["blas11"]["blases"][0]
    assert max(b["count"]) == 5 and by["blas11"]["midpoint_messages"] == 1  # the aborted split's leaf of five
    assert b["indices"][6:] == [5, 9, 3, 1, 7]  # ...whose indices the partition loop rotated before giving up
    assert b["count"][3:5] == [2, 4]  # the mirror-image cost tie went to the FIRST boundary (2 | 4, not 4 | 2)


def literal_statement():
    import importlib.util
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle", "derive_tree_fixtures.py")
    spec = importlib.util.spec_from_file_location("derive_tree_fixtures", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def irregular_scenes():
    """Inputs whose decisions DO hinge on roundings and ties: random soups (with exact duplicates), a regular grid
    (thousands of equal centroid coordinates, centroids on bin borders), a sliver-thin sheet, several meshes."""
    from raytracer_amd import meshgen
    soup, _ = meshgen.random_soup(1500, seed=21, extent=7.0, size=0.9)
    soup[0, 200:260] = soup[0, :60]  # exact duplicates: cost ties
    soup[0, 300:306] = soup[0, 0]  # eight copies of one triangle: an unsplittable clump (extent 0 -> midpoint -> abort)
    grid, _ = meshgen.plate(24)  # 1152 triangles on a regular grid
    thin, _ = meshgen.random_soup(700, seed=5, extent=3.0, size=0.4)
    thin[0, :, :, 2] *= 1e-9  # nearly planar: tiny extents on one axis
    thin[0, :, :, 0] += 11.0
    far, _ = meshgen.plate(9, x0=-14.0, y0=6.0)
    s2, _ = meshgen.random_soup(400, seed=77, extent=2.0, size=0.3)
    s2[0, :, :, 1] -= 9.0
    yield "soup", [soup]
    yield "grid", [grid]
    yield "five_meshes", [soup, grid, thin, far, s2]


@pytest.mark.parametrize("name,meshes", list(irregular_scenes()), ids=[n for n, _ in irregular_scenes()])
def test_both_builders_equal_the_literal_statement_on_irregular_meshes(name, meshes):
    """The fixtures above are chosen so that no decision hinges on a rounding.  Here the same line-by-line statement
    of rtbvh.cpp (oracle/derive_tree_fixtures.py, binary64 only) is run on irregular meshes of a few thousand
    triangles, where bin indices, cost comparisons and `centroid < split` do fall on roundings and ties: the oracle's
    builder and the product's (1 and 4 threads, and the frame driver's one-thread-per-tree mode) must produce its
    trees node for node -- boxes bit for bit, sorted order, slot order, message count."""
    lit = literal_statement().build_scene_binary64([m[0] for m in meshes])
    coords = [np.ascontiguousarray(m[:1]) for m in meshes]
    colors = [np.full((1, c.shape[1], 3), 0.5) for c in coords]

    def nodes_of(nodes):
        return dict(boxes=[n["box"].six() for n in nodes], count=[n["count"] for n in nodes],
                    left=[n["left"] for n in nodes], min_idx=[n["min_idx"] for n in nodes])

    want_tlas = nodes_of(lit["tlas"])
    builders = [("oracle", lambda: orc.Scene(coords, colors))]
    for nt in (1, 4, -1):
        builders.append((f"product n_threads={nt}", lambda nt=nt: ra.HostScene(coords, colors, 0, n_threads=nt)))
    for who, make in builders:
        sc = make()
        t = sc.tlas()
        check_nodes(dict(boxes=t["boxes"], left=t["left"], count=t["blas_count"], min_idx=t["min_blas_idx"]), want_tlas,
                    f"{who}: TLAS")
        msgs = sc.midpoint_msgs if who == "oracle" else sc.info.midpoint_fallbacks
        assert msgs == lit["messages"], who
        for s, mesh in enumerate(lit["slots"]):
            got = sc.slot(s)
            assert got["mesh"] == mesh, f"{who}: slot {s}"
            want = lit["blases"][mesh]
            check_nodes(got, nodes_of(want["nodes"]), f"{who}: slot {s} = mesh {mesh}")
            assert list(got["indices"]) == want["indices"], f"{who}: sorted order of mesh {mesh}"
    if name == "soup":
        assert lit["messages"] >= 1  # the duplicates end in a midpoint fallback somewhere


@pytest.mark.gpu
@pytest.mark.parametrize("fx", FIXTURES, ids=[f["name"] for f in FIXTURES])
def test_gpu_walk_reports_the_hand_derived_hits(fx):
    coords, colors = arrays(fx)
    o, d, mesh, elem, t = probe_arrays(fx)
    host = ra.HostScene(coords, colors, 0)
    ctx = ra.Context(0)
    try:
        ctx.upload(host)
        got = ctx.trace(o, d)
        gm, ge = host.map_prims(got["prim"])
        assert np.array_equal(gm, mesh) and np.array_equal(ge, elem)
        want = orc.Scene(coords, colors).trace(o, d)
        assert np.array_equal(got["t"].view(np.uint64), want["t"].view(np.uint64))
        # a camera above the scene looking down: the packet kernel with its cuts, not only trace_kernel
        from raytracer_amd import meshgen
        lo = np.min([c.reshape(-1, 3).min(0) for c in coords], 0)
        hi = np.max([c.reshape(-1, 3).max(0) for c in coords], 0)
        mid = (lo + hi) / 2
        H, W, spp = 64, 96, 4
        cam = meshgen.camera_arrays(W, H, center=(mid[0], mid[1], 0.6 * max(hi[0] - lo[0], hi[1] - lo[1]) + 5.0),
                                    target=(mid[0], mid[1], 0.0), fov=90)
        ctx.rng_seed(5489)
        ctx.render(*cam, H, W, spp, hit_ids=True, hit_t=True)
        g = ctx.download()
        w = orc.Scene(coords, colors).render(*cam, H, W, spp, rng=orc.Rng())
        gm, ge = host.map_prims(g["hit_prim"])
        assert np.array_equal(gm, w["hit_mesh"]) and np.array_equal(ge, w["hit_elem"])
        assert np.array_equal(g["hit_t"].view(np.uint64), w["hit_t"].view(np.uint64))
        assert np.array_equal(g["intensity"].view(np.uint64), w["intensity"].view(np.uint64))
        assert (w["hit_mesh"] >= 0).any()
    finally:
        ctx.close()
