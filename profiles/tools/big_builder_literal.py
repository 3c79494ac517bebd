"""Two meshes above the builder's chunk-parallel threshold (65 536 elements: parallel binning, closed-form partition,
task-split top) against the line-by-line statement of rtbvh.cpp in binary64 (oracle/derive_tree_fixtures.py; minutes of
pure Python per mesh): nodes, box bits, sorted order, message count.  CPU only; the statement is the checker."""
import sys, time, numpy as np
import os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import orc, raytracer_amd as ra
from raytracer_amd import meshgen
import test_tree_fixtures as tf
lit_mod = tf.literal_statement()
def nodes_of(ns):
    return dict(boxes=[n["box"].six() for n in ns], count=[n["count"] for n in ns], left=[n["left"] for n in ns], min_idx=[n["min_idx"] for n in ns])
for name, c in (("plate 190 (72200 triangles, row order)", meshgen.plate(190, side=5.0, amp=0.2)[0]),
                ("soup 70000", meshgen.random_soup(70000, seed=4, extent=6.0, size=0.2)[0])):
    c = np.ascontiguousarray(c[:1]); k = np.full((1, c.shape[1], 3), 0.5)
    t0 = time.time(); lit = lit_mod.build_scene_binary64([c[0]]); t1 = time.time()
    want = nodes_of(lit["blases"][0]["nodes"])
    for nt in (0, 1, -1):
        sc = ra.HostScene([c], [k], 0, n_threads=nt)
        got = sc.slot(0)
        tf.check_nodes(got, want, name)
        assert np.array_equal(np.ascontiguousarray(got["boxes"]).view(np.uint64), np.array(want["boxes"]).view(np.uint64))
        assert list(got["indices"]) == lit["blases"][0]["indices"]
        assert sc.info.midpoint_fallbacks == lit["messages"]
    print(f"{name}: product (pool, 1 thread, one thread per tree) == literal statement, {len(want['left'])} nodes, literal {t1 - t0:.0f} s", flush=True)
