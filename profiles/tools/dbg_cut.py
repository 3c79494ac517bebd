import os, sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import raytracer_amd as ra
from raytracer_amd import meshgen
T, H, W, spp = 3, 20, 32, 5
c0, k0 = meshgen.plate(10, T=T, sway=0.15)
c1, k1 = meshgen.uv_sphere(6, 8, radius=0.8, center=(0.4, 0.3, -2.5), T=T)
cams = [meshgen.camera_arrays(W, H), meshgen.camera_arrays(W, H, center=(0.3, 0.1, 0.2), target=(0, 0, -2), fov=70)]
ctx = ra.Context(0)
for t in range(T):
    host = ra.HostScene([c0, c1], [k0, k1], t)
    ctx.upload(host)
    for ci, cam in enumerate(cams):
        res = {}
        for mode in ("cut", "nocut"):
            if mode == "nocut": os.environ["RT_NO_CUT"] = "1"
            else: os.environ.pop("RT_NO_CUT", None)
            ctx.rng_seed(5489)
            ctx.render(*cam, H, W, spp, hit_ids=True, hit_t=True)
            res[mode] = ctx.download()
        a, b = res["cut"], res["nocut"]
        bad = np.nonzero(a["hit_prim"] != b["hit_prim"])[0]
        print("t", t, "cam", ci, "mismatching samples", bad.size, "of", a["hit_prim"].size)
        for s in bad[:10]:
            pix, k = divmod(int(s), spp); py, px = divmod(pix, W)
            pm, pe = host.map_prims(np.array([a["hit_prim"][s], b["hit_prim"][s]]))
            print("   px", px, "py", py, "k", k, "cut prim", a["hit_prim"][s], (pm[0], pe[0]), "t", a["hit_t"][s], "| nocut", b["hit_prim"][s], (pm[1], pe[1]), "t", b["hit_t"][s])

# the cuts of the failing frame
host = ra.HostScene([c0, c1], [k0, k1], 1)
ctx.upload(host)
os.environ.pop("RT_NO_CUT", None)
ctx.rng_seed(5489)
ctx.render(*cams[1], H, W, spp, hit_ids=True)
words, tx, tl = ctx.last_cuts()
print("tiles", words.shape, tx, tl)
for t in range(words.shape[0]):
    n = words[t, 0] & 0xff
    print(t, hex(words[t, 0]), [hex(x) for x in words[t, 1:1 + n]])
for s in range(2):
    sl = host.slot(s)
    print("slot", s, "mesh", sl["mesh"], "tri_base", sl["tri_base"], "nodes", len(sl["left"]))
