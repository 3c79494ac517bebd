"""Host BVH build (raytracer_amd/csrc/rt_bvh.cpp, through the C ABI) against the oracle's independent
restatement of cpp/rtbvh.cpp: node-for-node equality of every BLAS, the TLAS, the stored slot order and
the sorted element permutation.  CPU only."""
import numpy as np
import pytest

import orc
import raytracer_amd as ra
from raytracer_amd import meshgen


def assert_same_tree(coords, colors, timestep=0):
    o = orc.Scene(coords, colors, timestep)
    p = ra.HostScene(coords, colors, timestep)
    assert p.info.n_meshes == o.n_slots == len(coords)
    assert p.info.midpoint_fallbacks == o.midpoint_msgs
    to, tp = o.tlas(), p.tlas()
    for k in ("blas_count", "left", "min_blas_idx"):
        assert np.array_equal(to[k], tp[k]), k
    assert np.array_equal(to["boxes"].view(np.uint64), tp["boxes"].view(np.uint64))  # bit for bit (sign of zero too)
    total = 0
    for s in range(o.n_slots):
        so, sp = o.slot(s), p.slot(s)
        assert so["mesh"] == sp["mesh"]
        for k in ("left", "count", "min_idx", "indices"):
            assert np.array_equal(so[k], sp[k]), (s, k)
        assert np.array_equal(so["boxes"].view(np.uint64), sp["boxes"].view(np.uint64)), s  # bit for bit
        assert sp["tri_base"] == total
        total += sp["indices"].size
        # prim tables agree with the permutation
        pm, pe = p.prim_tables()
        assert np.all(pm[sp["tri_base"]:sp["tri_base"] + sp["indices"].size] == sp["mesh"])
        assert np.array_equal(pe[sp["tri_base"]:sp["tri_base"] + sp["indices"].size], sp["indices"])
    return o, p


def test_single_plate():
    c, k = meshgen.plate(40)
    o, p = assert_same_tree([c], [k])
    assert p.info.n_tris == 3200
    s = p.slot(0)
    leaves = s["left"] == -1
    assert s["count"][leaves].sum() == 3200 and (s["count"][~leaves] == 0).all()
    assert s["count"][leaves].max() <= 4
    # children adjacent, allocated after the parent
    inner = np.where(~leaves)[0]
    assert (s["left"][inner] > inner).all()


def test_tiny_meshes_are_single_leaves():
    c, k = meshgen.random_soup(3, seed=1)
    o, p = assert_same_tree([c], [k])
    assert p.slot(0)["left"].tolist() == [-1]
    c1, k1 = meshgen.random_soup(1, seed=2)
    assert_same_tree([c1], [k1])


@pytest.mark.parametrize("n_meshes", [1, 2, 3, 4, 5, 8, 9])
def test_multi_mesh_tlas_and_slot_order(n_meshes):
    rng = np.random.default_rng(100 + n_meshes)
    coords, colors = [], []
    for m in range(n_meshes):
        c, k = meshgen.random_soup(int(rng.integers(5, 400)), seed=1000 + m, extent=1.0)
        c = c + rng.uniform(-8, 8, 3)
        coords.append(c)
        colors.append(k)
    o, p = assert_same_tree(coords, colors)
    # every mesh appears in exactly one slot
    assert sorted(p.slot(s)["mesh"] for s in range(n_meshes)) == list(range(n_meshes))


def test_slot_order_differs_from_sorted_order_somewhere():
    """Quirk 1 (rtbvh.cpp:544-561 vs rtrayintersection.cpp:214-218) must be reproduced, not fixed:
    find a scene where slot s does not hold the mesh at sorted position s."""
    found = False
    for seed in range(40):
        rng = np.random.default_rng(seed)
        coords, colors = [], []
        for m in range(5):
            c, k = meshgen.random_soup(6, seed=seed * 10 + m, extent=0.5)
            coords.append(c + rng.uniform(-10, 10, 3))
            colors.append(k)
        o = orc.Scene(coords, colors)
        p = ra.HostScene(coords, colors)
        t = p.tlas()
        leaves = np.where(t["left"] == -1)[0]
        # sorted positions covered by leaves in node-index order
        order = np.concatenate([np.arange(t["min_blas_idx"][i], t["min_blas_idx"][i] + t["blas_count"][i])
                                for i in leaves])
        for s in range(5):
            assert o.slot(s)["mesh"] == p.slot(s)["mesh"]
        if not np.array_equal(order, np.arange(5)):
            found = True
    assert found


def test_degenerate_inputs_midpoint_fallback_and_big_leaves():
    # all centroids coincide: extent 0 -> midpoint message, every triangle goes right, split fails -> one big leaf
    base = np.array([[0.0, 0, -3], [1, 0, -3], [0, 1, -3]])
    c = np.repeat(base[None], 9, axis=0)[None]
    k = np.full((1, 9, 3), 0.25)
    o, p = assert_same_tree([c], [k])
    assert p.info.midpoint_fallbacks == 1
    assert p.info.max_leaf_tris == 9
    # collinear centroids along x with duplicates
    tris = []
    for i in range(40):
        x = float(i // 4)
        tris.append([[x, 0, -3], [x + 1, 0, -3], [x, 1, -3]])
    c = np.array(tris)[None]
    k = np.full((1, 40, 3), 0.5)
    assert_same_tree([c], [k])


def test_timestep_selection():
    c, k = meshgen.plate(12, T=4, sway=0.1)
    for t in range(4):
        assert_same_tree([c], [k], timestep=t)


def test_thread_count_does_not_change_the_tree():
    cs, ks = meshgen.plate_grid(16, grid=(2, 2))
    a = ra.HostScene(cs, ks, n_threads=1)
    b = ra.HostScene(cs, ks, n_threads=4)
    for s in range(4):
        sa, sb = a.slot(s), b.slot(s)
        for key in sa:
            assert np.array_equal(sa[key], sb[key])


def test_one_thread_per_tree_mode_with_builds_in_flight_gives_the_same_trees():
    """n_threads = -1 (include/rtgpu.h): the frame driver's way of building -- every tree by one thread of the pool,
    several builds in flight on the caller's threads.  Same trees as the task-split build, also for meshes large
    enough to take the task-split path otherwise (>= 8192 triangles)."""
    from concurrent.futures import ThreadPoolExecutor
    cs, ks = [], []
    for m in range(5):
        c, k = meshgen.plate(72 + 3 * m, x0=-8.0 + 3.5 * m, y0=-1.0 + 0.3 * m, T=3, sway=0.05)  # 10,368 .. 14,112 triangles
        cs.append(np.ascontiguousarray(c))
        ks.append(np.ascontiguousarray(k))
    want = [ra.HostScene(cs, ks, t, n_threads=1) for t in range(3)]
    with ThreadPoolExecutor(6) as ex:
        got = list(ex.map(lambda j: ra.HostScene(cs, ks, j % 3, n_threads=-1), range(12)))
    for j, g in enumerate(got):
        w = want[j % 3]
        tw, tg = w.tlas(), g.tlas()
        for key in tw:
            assert np.array_equal(tw[key], tg[key]), key
        for s in range(5):
            sw, sg = w.slot(s), g.slot(s)
            for key in sw:
                assert np.array_equal(sw[key], sg[key]), (j, s, key)


def test_rejects_bad_input():
    c, k = meshgen.plate(4)
    with pytest.raises(ValueError):
        ra.HostScene([c], [k[:, :-1]])
    with pytest.raises(ValueError):
        ra.HostScene([c], [k], timestep=3)
    with pytest.raises(ra.RtError):
        ra.HostScene([np.zeros((1, 0, 3, 3))], [np.zeros((1, 0, 3))])


def test_stack_need_covers_depth():
    c, k = meshgen.plate(64)
    p = ra.HostScene([c], [k])
    s = p.slot(0)
    # depth of the tree by walking parents
    depth = np.zeros(s["left"].size, int)
    for i in np.where(s["left"] >= 0)[0]:
        depth[s["left"][i]] = depth[i] + 1
        depth[s["left"][i] + 1] = depth[i] + 1
    assert 2 <= p.info.stack_need <= depth.max() + 3


def test_large_meshes_take_the_parallel_build_and_match_the_serial_oracle():
    """>= 131072 triangles: the top of the tree is expanded with chunk-parallel scans and the subtrees are
    built concurrently and spliced (rt_bvh.cpp build_tree); the result must still equal the oracle's
    serial build node for node -- boxes bit for bit, permutation, numbering, fallback count."""
    c, k = meshgen.plate(300)  # 180,000 triangles, regular grid (many equal centroid coordinates)
    o, p = assert_same_tree([c], [k])
    assert p.info.n_tris == 180000
    # irregular soup with duplicates (ties in the partition, degenerate bins) next to a second big mesh
    c2, k2 = meshgen.random_soup(150000, seed=9, extent=40.0, size=0.4)
    c2[0, 1000:2000] = c2[0, :1000]
    c3, k3 = meshgen.plate(260, x0=50.0, y0=-3.0)
    assert_same_tree([c2, c3], [k2, k3])


def test_signed_zero_extremes_follow_the_serial_scan_order():
    """Box extremes are exact min/max, so only the SIGN of a zero extreme can depend on the order in which
    triangles are visited (+0 == -0).  The builder gathers child boxes while it partitions (another order than
    the reference's range scan) and must fall back to the ordered scan for zero extremes: meshes full of +0/-0
    coordinates on box faces must still match the oracle's serial build bit for bit."""
    rng = np.random.default_rng(12)
    for n in (20, 150):
        c, k = meshgen.plate(n, side=8.0, amp=0.0, z0=0.0)   # flat sheet in the z = 0 plane, x/y grid lines at 0
        c = c.copy()
        flip = rng.random(c.shape) < 0.5
        c[(c == 0.0) & flip] = -0.0
        assert (np.signbit(c) & (c == 0.0)).any() and (~np.signbit(c) & (c == 0.0)).any()
        assert_same_tree([c], [k])


def test_parallel_partition_is_the_serial_two_pointer_permutation():
    """Ranges of >= 65,536 elements are partitioned by a chunk-parallel closed form of the reference's sequential
    loop (cpp/rtbvh.cpp:271-285; rt_bvh.cpp partition_closed_form).  The permutation must be the loop's own, element for
    element -- leaves keep this order and ties in t go to the first -- on 10^6 random class patterns with random chunk
    lengths (all-left, all-right, holes and fillers on chunk borders included), and on longer ranges."""
    from raytracer_amd import _lib as L
    assert L.lib.rt_partition_selftest(1_000_000, 12345, 40) == 0
    assert L.lib.rt_partition_selftest(20_000, 777, 600) == 0
    assert L.lib.rt_partition_selftest(200, 99, 200_000) == 0


def test_signed_zero_coordinates_build_the_same_tree():
    """std::min / std::max keep the FIRST of two equal zeros (rtbvh.h AABB::expand_*): the builder's SSE2 box updates
    (record init on registers, three 16-byte updates per box, bins, sides from bin unions) must reproduce that, bit for
    bit, on coordinates full of +0.0 and -0.0."""
    rng = np.random.default_rng(77)
    for trial in range(6):
        c, k = meshgen.random_soup(int(rng.integers(40, 900)), seed=500 + trial, extent=2.0, size=0.7)
        flat = c.reshape(-1)
        n = flat.size
        flat[rng.integers(0, n, n // 6)] = 0.0
        flat[rng.integers(0, n, n // 6)] = -0.0
        assert_same_tree([c], [k])
        c2, k2 = meshgen.random_soup(30, seed=900 + trial, extent=1.0)
        assert_same_tree([c, c2 + 3.0], [k, k2])


def test_non_finite_centroids_are_refused_where_the_reference_indexes_its_bins_with_them():
    """A NaN or infinite centroid makes `static_cast<int>(t * NUM_BINS)` (rtbvh.cpp:184) undefined behaviour as soon as
    the mesh is split (more than 4 triangles; the TLAS: more than 2 meshes): an error, not a crash.  Meshes the reference
    never splits are built as they are, NaNs and all (std::min / std::max skip an incoming NaN)."""
    from raytracer_amd._lib import RtError
    for bad in (np.nan, np.inf, -np.inf, 1.7e308):
        c, k = meshgen.random_soup(200, seed=5, extent=2.0, size=0.7)
        c.reshape(-1)[37] = bad
        if bad == 1.7e308:
            c.reshape(-1)[40] = bad  # the same component of the next vertex: the centroid's sum overflows
        with pytest.raises(RtError, match="not finite"):
            ra.HostScene([c], [k], 0)
        c4, k4 = meshgen.random_soup(4, seed=6)
        c4.reshape(-1)[5] = bad
        if np.isnan(bad):
            assert_same_tree([c4], [k4])
    many = [meshgen.random_soup(3, seed=10 + m) for m in range(3)]
    many[1][0].reshape(-1)[2] = np.inf
    with pytest.raises(RtError, match="not finite"):
        ra.HostScene([m[0] for m in many], [m[1] for m in many], 0)
    assert_same_tree([m[0] for m in many[:2]], [m[1] for m in many[:2]])  # two meshes: one TLAS leaf, no split
