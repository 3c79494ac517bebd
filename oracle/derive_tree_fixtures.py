#!/usr/bin/env python3
"""Hand-derivable tree fixtures -- TEST INFRASTRUCTURE ONLY (writes tests/golden/tree_fixtures.json and
tests/golden/tree_fixtures_derivation.txt; run it from the repository root, it needs nothing but Python).

Why this exists.  The reference's builder and walk (cpp/rtbvh.cpp, cpp/rtrayintersection.cpp) cannot be compiled
here (they include <nanobind/...>), so the oracle's builder (oracle/rt_oracle.c) and the product's
(raytracer_amd/csrc/rt_bvh.cpp) were only ever compared with each other.  This script is a THIRD statement of
build_acceleration_structures, deliberately naive, one reference line at a time (each step cites the line it
follows), run on inputs small enough to follow with pencil and paper, and it writes every intermediate value --
centroid bounds, axis, bin of every element, prefix/suffix counts and boxes, the cost of every bin boundary, the
split position, every swap of the partition loop -- to the derivation file, so that a reader can check the expected
trees against the reference source without trusting any of the three implementations.

Every quantity is computed twice: in IEEE binary64 in the reference's operation order (Python floats ARE binary64,
`/` and `*` round like C) and in exact rational arithmetic (fractions.Fraction).  Every DECISION -- a comparison
that steers the build: longest axis, bin index, lower cost, centroid < split position -- must come out the same in
both, and where two costs tie they must tie exactly in both; the script stops otherwise.  The expected trees are
therefore what the reference's algorithm yields in exact arithmetic too: they do not hinge on a rounding.

Inputs are chosen to make the build take the branches nothing else reference-derived pins:
  tlas5   5 meshes: the root splits 3|2, the LEFT child is split again, so the leaves do not appear in range order and
          copy_data_to_TLAS (rtbvh.cpp:552-560, node-index order) disagrees with intersect_TLAS
          (rtrayintersection.cpp:214-218, sorted-range order): quirk 1.  Also a four-way cost tie at the second split.
  tlas8   8 meshes, a cluster of five and a column of three: splits on x, then y, then x again; leaves appear out of
          range order at two levels.
  blas11  one mesh of 11 triangles: a SAH cost tie between mirror-image boundaries at the first split inside the row
          of six (first strictly lower cost wins, rtbvh.cpp:231), and five triangles with one common centroid: centroid
          extent 0 -> midpoint fallback (rtbvh.cpp:169-173) -> nothing is `< split` -> the partition loop rotates the
          five indices and leaves the left side empty -> split aborted, a leaf of five (rtbvh.cpp:295-297, 352-358).
"""
import json
import os
import sys
from fractions import Fraction as Fr

INF = float("inf")
LOG = []


def log(s=""):
    if EXACT:
        LOG.append(s)


def fr(x):
    return Fr(x) if x not in (INF, -INF) else x


def show(x):
    """exact value of a float for the derivation file"""
    if x in (INF, -INF):
        return "inf" if x > 0 else "-inf"
    f = Fr(x)
    return str(f.numerator) if f.denominator == 1 else f"{f.numerator}/{f.denominator}"


EXACT = True  # False (build_scene_binary64): binary64 only, decisions may hinge on a rounding, nothing is logged


class Both:
    """A value carried in binary64 (reference order of operations) and, when EXACT, as an exact rational."""
    __slots__ = ("f", "q")

    def __init__(self, f, q=None):
        self.f = float(f)
        self.q = (Fr(f) if q is None else q) if EXACT else None

    def __add__(self, o):
        return Both(self.f + o.f, self.q + o.q if EXACT else None)

    def __sub__(self, o):
        return Both(self.f - o.f, self.q - o.q if EXACT else None)

    def __mul__(self, o):
        return Both(self.f * o.f, self.q * o.q if EXACT else None)

    def __truediv__(self, o):
        return Both(self.f / o.f, self.q / o.q if EXACT else None)

    def lt(self, o, what):
        if not EXACT:
            return self.f < o.f
        a, b = self.f < o.f, self.q < o.q
        if a != b:
            sys.exit(f"decision '{what}' depends on rounding: float {self.f!r} < {o.f!r} is {a}, exact is {b}")
        if (self.f == o.f) != (self.q == o.q):
            sys.exit(f"tie in '{what}' is not exact in both arithmetics: {self.f!r} vs {o.f!r}")
        return a

    def same(self, o):  # equal values (ties)
        return self.q == o.q if EXACT else self.f == o.f

    def __str__(self):
        if not EXACT:
            return repr(self.f)
        q = self.q
        return str(q.numerator) if q.denominator == 1 else f"{q.numerator}/{q.denominator} (={self.f!r})"


def B(x):
    return Both(x)


class AABB:  # rtbvh.h:33-81
    def __init__(self):  # :37-40
        self.mn = [INF, INF, INF]
        self.mx = [-INF, -INF, -INF]

    def add_point(self, p):  # :55-61  std::min(a,b) = (b<a)?b:a ; std::max(a,b) = (a<b)?b:a
        for i in range(3):
            self.mn[i] = p[i] if p[i] < self.mn[i] else self.mn[i]
            self.mx[i] = p[i] if self.mx[i] < p[i] else self.mx[i]

    def add_box(self, o):  # :63-68
        for i in range(3):
            self.mn[i] = o.mn[i] if o.mn[i] < self.mn[i] else self.mn[i]
            self.mx[i] = o.mx[i] if self.mx[i] < o.mx[i] else self.mx[i]

    def extent(self, a):  # :70-74   (inputs are finite here, so the subtraction is exact in both arithmetics)
        r = B(self.mx[a]) - B(self.mn[a])
        return B(0.0) if r.f < 0 else r

    def area(self):  # :75-80
        h, w, d = self.extent(2), self.extent(1), self.extent(0)
        return B(2.0) * (h * w + w * d + h * d)

    def copy(self):
        c = AABB()
        c.mn, c.mx = list(self.mn), list(self.mx)
        return c

    def six(self):
        return self.mn + self.mx

    def __str__(self):
        return "[" + ", ".join(show(v) for v in self.mn) + "]..[" + ", ".join(show(v) for v in self.mx) + "]"


def node_aabb(boxes, idx, lo, n):  # create_node_AABB, rtbvh.cpp:81-94
    b = AABB()
    for i in range(lo, lo + n):
        b.add_box(boxes[idx[i]])
    return b


NUM_BINS = 8  # rtbvh.cpp:176
FLAGS = set()


def binned_sah_split(count, lo, cent, boxes, idx, msgs):  # rtbvh.cpp:132-247 -> (found, axis, position)
    if count <= 2:  # :140
        return False, 0, None
    cb = AABB()  # :147-153 centroid bounds
    for i in range(lo, lo + count):
        cb.add_point([c.f for c in cent[idx[i]]])
    ext = [cb.extent(a) for a in range(3)]
    axis, e = 0, ext[0]  # :156-164 longest axis, first maximum wins (strict >)
    for a in (1, 2):
        if e.lt(ext[a], "longest axis"):
            axis, e = a, ext[a]
    log(f"    centroid bounds {cb}; extents x={ext[0]} y={ext[1]} z={ext[2]} -> axis {'xyz'[axis]}, extent {e}")
    if e.f == 0:  # :169-173 -> midpoint_split :122-130
        assert not EXACT or e.q == 0
        msgs.append("SAH splitting failed. Trying midpoint instead.")
        pos = B(cb.mn[axis]) + e * B(0.5)
        FLAGS.add("extent0_midpoint")
        log(f"    extent == 0 -> midpoint fallback (rtbvh.cpp:169-173, prints the message): position = {pos}")
        return True, axis, pos
    bins_n = [0] * NUM_BINS
    bins_b = [AABB() for _ in range(NUM_BINS)]
    inv = B(1.0) / e  # :179
    for i in range(lo, lo + count):  # :180-188
        el = idx[i]
        t = (cent[el][axis] - B(cb.mn[axis])) * inv
        tb = t * B(float(NUM_BINS))
        bf = int(tb.f)  # static_cast<int>: truncation (values are >= 0)
        if EXACT and bf != int(tb.q):
            sys.exit(f"bin of element {el} depends on rounding ({tb.f!r} vs {tb.q})")
        b = NUM_BINS - 1 if bf == NUM_BINS else bf  # :185
        bins_n[b] += 1
        bins_b[b].add_box(boxes[el])
        log(f"      element {el}: centroid[{'xyz'[axis]}] = {cent[el][axis]} -> t*8 = {tb} -> bin {b}")
    lcnt, rcnt, lbox, rbox = [0] * NUM_BINS, [0] * NUM_BINS, [None] * NUM_BINS, [None] * NUM_BINS
    run, n = AABB(), 0
    for i in range(NUM_BINS):  # :195-204
        if bins_n[i] > 0:
            run.add_box(bins_b[i])
        n += bins_n[i]
        lbox[i], lcnt[i] = run.copy(), n
    run, n = AABB(), 0
    for i in range(NUM_BINS - 1, -1, -1):  # :206-216
        if bins_n[i] > 0:
            run.add_box(bins_b[i])
        n += bins_n[i]
        rbox[i], rcnt[i] = run.copy(), n
    parent = node_aabb(boxes, idx, lo, count)  # :221
    ap = parent.area()
    log(f"    bin counts {bins_n}; parent box {parent}, area {ap}")
    best_cost, best = None, -1
    costs = []
    for i in range(NUM_BINS - 1):  # :223-235
        nl, nr = lcnt[i], rcnt[i + 1]
        if nl == 0 or nr == 0:
            continue
        al, ar = lbox[i].area(), rbox[i + 1].area()
        # find_SAH_cost_bin_full :108-120:  1.0 + 2.0 * (al/ap * nl + ar/ap * nr)
        cost = B(1.0) + B(2.0) * (al / ap * B(float(nl)) + ar / ap * B(float(nr)))
        costs.append((i, cost))
        lower = best_cost is None or cost.lt(best_cost, f"cost of boundary {i} < best")
        if best_cost is not None and cost.same(best_cost) and best != -1:
            FLAGS.add("cost_tie_with_best")
        log(f"      boundary after bin {i}: left n={nl} area {al}, right n={nr} area {ar} -> cost {cost}"
            + ("  <- new best" if lower else ""))
        if lower:  # :231 strictly lower: the FIRST of equal costs wins
            best_cost, best = cost, i
    if best == -1:  # :236-240
        msgs.append("SAH splitting failed. Trying midpoint instead.")
        pos = B(cb.mn[axis]) + e * B(0.5)
        FLAGS.add("no_valid_bin_midpoint")
        return True, axis, pos
    ties = [i for i, c in costs if c.same(best_cost)]
    if len(ties) > 1:
        log(f"    boundaries {ties} tie at the lowest cost {best_cost}: the first, {best}, wins (rtbvh.cpp:231, strict <)")
    width = e / B(float(NUM_BINS))  # :243
    pos = B(cb.mn[axis]) + width * B(float(best + 1))  # :244
    log(f"    best boundary: after bin {best}; split position = {pos}")
    return True, axis, pos


def split_node(count, lo, cent, boxes, idx, msgs):  # split_BVH_node, rtbvh.cpp:249-299 -> (ok, left_count)
    found, axis, pos = binned_sah_split(count, lo, cent, boxes, idx, msgs)
    if not found:  # :265-268
        return False, 0
    begin, end, mid = lo, lo + count, lo  # :271-273
    while mid < end:  # :275-285 two-pointer partition, unstable
        el = idx[mid]
        if cent[el][axis].lt(pos, f"centroid of {el} < split"):
            mid += 1
        else:
            end -= 1
            idx[mid], idx[end] = idx[end], idx[mid]
            log(f"      partition: element {el} is not left of the split -> swap positions {mid},{end}: "
                f"indices[{lo}:{lo + count}] = {idx[lo:lo + count]}")
    left = mid - begin  # :287
    log(f"    partition done: indices[{lo}:{lo + count}] = {idx[lo:lo + count]}, left {left}, right {count - left}")
    if left == 0 or count - left == 0:  # :295-297
        FLAGS.add("empty_side_abort")
        log("    one side is empty -> split aborted, the node stays a leaf (rtbvh.cpp:295-297); "
            "the indices keep the order the loop left them in")
        return False, left
    return True, left


def build_tree(n_elem, cent, boxes, idx, max_leaf, msgs, what):
    """build_BLAS rtbvh.cpp:301-390 (max_leaf 4) / build_TLAS :392-476 (max_leaf 2): the same loop.
    -> nodes: dicts box / count / left / min_idx."""
    nodes = [dict(box=node_aabb(boxes, idx, 0, n_elem), count=n_elem, left=-1, min_idx=0)]  # :317-323 / :404-408
    stack = [(n_elem, 0, 0)]  # (element_count, node_idx, min_element_idx)
    while stack:
        count, ni, lo = stack.pop()  # :332-336 LIFO
        log(f"  pop {what} node {ni}: elements indices[{lo}:{lo + count}] = {idx[lo:lo + count]}")
        if count <= max_leaf:  # :340-346 / :424-431
            nodes[ni].update(count=count, box=node_aabb(boxes, idx, lo, count), left=-1, min_idx=lo)
            log(f"    {count} <= {max_leaf}: leaf, box {nodes[ni]['box']}")
            continue
        ok, left_count = split_node(count, lo, cent, boxes, idx, msgs)
        if not ok:  # :352-359 / :437-445
            nodes[ni].update(count=count, box=node_aabb(boxes, idx, lo, count), left=-1, min_idx=lo)
            log(f"    leaf of {count} (> {max_leaf}), box {nodes[ni]['box']}")
            continue
        lc = len(nodes)  # :362-363 children are the next two nodes
        right_count, right_lo = count - left_count, lo + left_count
        nodes.append(dict(box=node_aabb(boxes, idx, lo, left_count), count=left_count, left=-1, min_idx=lo))
        nodes.append(dict(box=node_aabb(boxes, idx, right_lo, right_count), count=right_count, left=-1,
                          min_idx=right_lo))
        nodes[ni]["left"], nodes[ni]["count"] = lc, 0  # :383-384 / :469-470
        log(f"    -> internal; children {lc} (indices[{lo}:{lo + left_count}]) and {lc + 1} "
            f"(indices[{right_lo}:{right_lo + right_count}])")
        stack.append((right_count, lc + 1, right_lo))  # :387-388 right pushed first: left is popped first
        stack.append((left_count, lc, lo))
    return nodes


def build_scene(meshes):
    """build_acceleration_structures, rtbvh.cpp:590-685, timestep 0.  meshes: list of lists of 3x3 triangles."""
    msgs = []
    mesh_box, mesh_cent, blases = [], [], []
    for m, tris in enumerate(meshes):
        log(f"mesh {m}: {len(tris)} triangle(s)")
        cent, boxes, mb = [], [], AABB()
        for e, t in enumerate(tris):  # process_element_data_tri3 :36-79
            # compute_element_centroid_tri3 rtbvh.h:153-160: (a + b + c) / 3.0, left to right
            c = [(B(t[0][i]) + B(t[1][i]) + B(t[2][i])) / B(3.0) for i in range(3)]
            for v in c:
                if EXACT and Fr(v.f) != v.q:
                    sys.exit("a centroid is not exactly representable: choose other coordinates")
            cent.append(c)
            b = AABB()
            for node in range(3):  # build_for_tri3 rtbvh.h:43-52
                b.add_point(list(t[node]))
            boxes.append(b)
            mb.add_box(b)
            log(f"  triangle {e}: centroid ({c[0]}, {c[1]}, {c[2]}), box {b}")
        mesh_box.append(mb)
        mesh_cent.append([(B(mb.mn[i]) + B(mb.mx[i])) / B(2.0) for i in range(3)])  # compute_mesh_centroid :162-167
        idx = list(range(len(tris)))  # :638-640
        nodes = build_tree(len(tris), cent, boxes, idx, 4, msgs, f"BLAS[{m}]")
        blases.append(dict(mesh=m, nodes=nodes, indices=idx))
    log("TLAS over the mesh boxes; element = mesh, centroid = middle of the mesh box (rtbvh.h:162-167):")
    for m in range(len(meshes)):
        log(f"  mesh {m}: box {mesh_box[m]}, centroid ({mesh_cent[m][0]}, {mesh_cent[m][1]}, {mesh_cent[m][2]})")
    bidx = list(range(len(meshes)))  # :670-672
    tnodes = build_tree(len(meshes), mesh_cent, mesh_box, bidx, 2, msgs, "TLAS")
    # copy_data_to_TLAS :544-561: for every TLAS node IN NODE-INDEX ORDER, its range of scene_blas_indices
    slots = []
    for i, n in enumerate(tnodes):
        for j in range(n["min_idx"], n["min_idx"] + n["count"]):
            slots.append(bidx[j])
    log(f"scene_blas_indices after the build: {bidx}")
    log(f"tlas.blases (slot -> mesh), appended in node-index order (rtbvh.cpp:552-560): {slots}")
    guards = {}
    for i, n in enumerate(tnodes):
        if n["left"] == -1:
            rng = list(range(n["min_idx"], n["min_idx"] + n["count"]))
            guards[i] = dict(box_of_meshes=[bidx[j] for j in rng], walks_meshes=[slots[j] for j in rng])
            same = guards[i]["box_of_meshes"] == guards[i]["walks_meshes"]
            log(f"  TLAS leaf {i}: its box bounds meshes {guards[i]['box_of_meshes']}, intersect_TLAS walks "
                f"blases[{rng[0]}..{rng[-1]}] = meshes {guards[i]['walks_meshes']} (rtrayintersection.cpp:214-218)"
                + ("" if same else "  <- quirk 1"))
    if any(g["box_of_meshes"] != g["walks_meshes"] for g in guards.values()):
        FLAGS.add("quirk1_active")
    return dict(tlas=tnodes, blas_indices=bidx, slots=slots, blases=blases, messages=len(msgs), guards=guards)


def build_scene_binary64(meshes):
    """The same line-by-line statement in binary64 alone, for inputs whose decisions MAY hinge on a rounding (irregular
    meshes of a few thousand triangles: tests/test_tree_fixtures.py compares both builders with it node for node).
    meshes: list of (E, 3, 3) nested lists / arrays.  -> build_scene's dict, boxes as six floats."""
    global EXACT
    keep = EXACT
    EXACT = False
    try:
        return build_scene([[[[float(x) for x in v] for v in t] for t in tris] for tris in meshes])
    finally:
        EXACT = keep


def export(name, meshes, built, probes):
    def nodes_out(nodes):
        return dict(boxes=[n["box"].six() for n in nodes], count=[n["count"] for n in nodes],
                    left=[n["left"] for n in nodes], min_idx=[n["min_idx"] for n in nodes])

    return dict(name=name, meshes=[[[list(map(float, v)) for v in t] for t in tris] for tris in meshes],
                tlas=nodes_out(built["tlas"]), scene_blas_indices=built["blas_indices"],
                slot_to_mesh=built["slots"],
                blases=[dict(mesh=b["mesh"], indices=b["indices"], **nodes_out(b["nodes"])) for b in built["blases"]],
                midpoint_messages=built["messages"], probes=probes)


# ---------------------------------------------------------------------------------------------------- inputs
def quad(dx, dy, z=0.0, n_tris=1, s=3.0):
    """One or two triangles of the square [dx,dx+s] x [dy,dy+s] at height z, counter-clockwise seen from +z
    (a ray going down, direction (0,0,-1), has det = +s*s > 1e-6: front face, rtrayintersection.cpp:84)."""
    a, b, c, d = (dx, dy, z), (dx + s, dy, z), (dx + s, dy + s, z), (dx, dy + s, z)
    return [(a, b, d), (b, c, d)][:n_tris]


def down_probe(x, y):
    """A ray from (x, y, 10) straight down.  Which mesh it reports is derivable by hand: the ray passes a box iff
    (x, y) is inside it (probes keep >= 1/4 away from every box side and triangle edge), so it reaches exactly the
    TLAS leaves whose box contains (x, y), in node order of the left-first walk, and in each it is tested against
    the meshes the leaf WALKS (quirk 1), not the meshes its box bounds."""
    return dict(origin=[x, y, 10.0], dir=[0.0, 0.0, -1.0])


def inside(b, x, y):
    return b.mn[0] < x < b.mx[0] and b.mn[1] < y < b.mx[1]


def tri_contains(t, x, y):
    def side(p, q):
        return (q[0] - p[0]) * (y - p[1]) - (q[1] - p[1]) * (x - p[0])

    s = [side(t[0], t[1]), side(t[1], t[2]), side(t[2], t[0])]
    assert all(abs(v) > 0.5 for v in s), "probe too close to an edge"
    return all(v > 0 for v in s)


def expected_probe_hit(meshes, built, x, y):
    """Left-first walk of the TLAS with the hand rule above; closest hit = highest z (all rays go down); strict <
    on t keeps the first of equal heights (rtrayintersection.cpp:177)."""
    best = None  # (z, mesh, elem)
    stack = [0]
    tn = built["tlas"]
    while stack:
        i = stack.pop()
        if not inside(tn[i]["box"], x, y):
            continue
        if tn[i]["left"] == -1:
            for j in range(tn[i]["min_idx"], tn[i]["min_idx"] + tn[i]["count"]):
                m = built["slots"][j]  # blases[j]: the quirk
                bl = built["blases"][m]
                bstack = [0]  # intersect_BLAS, rtrayintersection.cpp:146-195: the same left-first walk
                while bstack:
                    k = bstack.pop()
                    bn = bl["nodes"][k]
                    if not inside(bn["box"], x, y):
                        continue
                    if bn["left"] == -1:
                        for row in range(bn["count"]):  # rows in stored (sorted-index) order, first minimum wins
                            e = bl["indices"][bn["min_idx"] + row]
                            t = meshes[m][e]
                            if tri_contains(t, x, y) and (best is None or t[0][2] > best[0]):
                                best = (t[0][2], m, e)
                    else:
                        bstack.append(bn["left"] + 1)
                        bstack.append(bn["left"])
        else:
            stack.append(tn[i]["left"] + 1)
            stack.append(tn[i]["left"])
    return dict(mesh=-1, elem=-1, t=None) if best is None else dict(mesh=best[1], elem=best[2], t=10.0 - best[0])


def geometric_hit(meshes, x, y):
    """What a renderer WITHOUT quirk 1 would report (for the derivation file only)."""
    best = None
    for m, tris in enumerate(meshes):
        for e, t in enumerate(tris):
            if tri_contains(t, x, y) and (best is None or t[0][2] > best[0]):
                best = (t[0][2], m, e)
    return None if best is None else best[1]


def make(name, meshes, probe_xy, need):
    global FLAGS
    FLAGS = set()
    log("=" * 100)
    log(f"fixture {name}")
    log("=" * 100)
    built = build_scene(meshes)
    missing = set(need) - FLAGS
    if missing:
        sys.exit(f"{name}: the inputs do not take the branches {sorted(missing)}")
    probes = []
    n_quirk = 0
    for x, y in probe_xy:
        p = down_probe(x, y)
        p["expect"] = expected_probe_hit(meshes, built, x, y)
        g = geometric_hit(meshes, x, y)
        n_quirk += (g if g is not None else -1) != p["expect"]["mesh"]
        log(f"probe ({x}, {y}) straight down: geometrically over mesh {g}; the reference reports mesh "
            f"{p['expect']['mesh']} element {p['expect']['elem']} t {p['expect']['t']}")
        probes.append(p)
    log(f"branches taken: {sorted(FLAGS)}; {n_quirk} of {len(probes)} probes see something else than the geometry")
    return export(name, meshes, built, probes), n_quirk


def main():
    out = []
    # ---- tlas5: clusters {0,4,8} and {56,64} along x, given in shuffled order; meshes overlap in height so the
    #      wrongly guarded ones are sometimes still seen: mesh z = its number, boxes of 3 x 3
    xs5 = [56, 0, 64, 8, 4]
    m5 = [quad(float(x), 0.0, z=float(i) / 4, n_tris=1 + i % 2, s=6.0) for i, x in enumerate(xs5)]
    fx, nq = make("tlas5", m5, [(x + 1.0, 0.75) for x in xs5] + [(5.0, 0.5), (9.5, 0.75), (30.0, 1.0)],
                  ["quirk1_active", "cost_tie_with_best"])
    assert nq > 0
    out.append(fx)
    # ---- tlas8: a cluster of five on the left (three in the row y = 0, two at y = 64) and a column of three far to
    #      the right, shuffled: the root splits 5|3 on x, the five split 3|2 on y, the three of the row once more on x
    #      (nodes 5,6 come after leaf 4 although their range comes before it), the column splits on y
    pos8 = [(124, 32), (0, 0), (8, 64), (4, 0), (128, 64), (8, 0), (0, 64), (120, 0)]
    m8 = [quad(float(x), float(y), z=float(i) / 4, n_tris=1 + (i % 3 == 0), s=6.0) for i, (x, y) in enumerate(pos8)]
    fx, nq = make("tlas8", m8, [(x + 1.0, y + 0.75) for x, y in pos8] + [(5.0, 0.5), (9.0, 0.5), (121.0, 5.25)],
                  ["quirk1_active"])
    assert nq > 0
    out.append(fx)
    # ---- blas11: one mesh.  Six triangles in a row (centroid x offsets 0,4 | 28,36 | 60,64 of an extent of 64:
    #      mirror-symmetric) at y = 0, five nested triangles with the common centroid (33, 258, 0) far away in y.
    #      The nested triangles are coplanar and their sizes are 1 : 2 : 4 : 8 : 16, so a ray through several of them
    #      gets bit-identical t from each (every intermediate scales by a power of two): a genuine tie, resolved by the
    #      leaf's stored order, which the aborted partition has rotated.
    row = [quad(float(dx), 0.0, n_tris=1)[0] for dx in (0, 4, 28, 36, 60, 64)]
    nest = [((33.0 - a, 258.0 - a, 0.0), (33.0 + a, 258.0 - a, 0.0), (33.0, 258.0 + 2 * a, 0.0))
            for a in (1.0, 2.0, 4.0, 8.0, 16.0)]
    tris = [row[3], nest[0], row[0], nest[1], row[5], nest[2], row[1], nest[3], row[4], nest[4], row[2]]
    fx, _ = make("blas11", [tris], [(1.0, 0.75), (37.0, 0.75), (33.0, 258.0), (33.0, 264.5), (33.0, 254.25),
                                    (33.0, 251.0), (33.0, 244.0)],
                 ["cost_tie_with_best", "extent0_midpoint", "empty_side_abort"])
    assert fx["midpoint_messages"] == 1 and max(fx["blases"][0]["count"]) == 5
    out.append(fx)

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gold = os.path.join(root, "tests", "golden")
    with open(os.path.join(gold, "tree_fixtures.json"), "w") as f:
        json.dump(out, f, indent=1)
    with open(os.path.join(gold, "tree_fixtures_derivation.txt"), "w") as f:
        f.write("Derivation of tests/golden/tree_fixtures.json, written by oracle/derive_tree_fixtures.py.\n"
                "Every value is exact (integers or fractions); line numbers refer to the reference's cpp/rtbvh.cpp,\n"
                "cpp/rtbvh.h and cpp/rtrayintersection.cpp.\n\n" + "\n".join(LOG) + "\n")
    print(f"wrote {len(out)} fixtures, {len(LOG)} derivation lines")


if __name__ == "__main__":
    main()
