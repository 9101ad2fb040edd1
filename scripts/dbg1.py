import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, hpt
from hpt import scene_io, host_bvh
import oracle_lib as ol
from golden.make_golden import scene_rays
desc = scene_io.load_scene(os.path.join(ROOT, "tests/golden/scene.xml"))
osc = ol.OracleScene(ol.load(), desc)
ctx = hpt.Context(0)
sc = hpt.Scene(ctx, desc, host_bvh=lambda m: host_bvh.build_reference_bvh(m.verts, m.tris))
o, d = scene_rays(20000, seed=11)
for v in (0, 7):
    hpt.lib.hpt_debug_set_variant(v)
    g = sc.intersect(o, d); r = osc.intersect(o, d)
    for k in ("node", "prim", "t", "side"):
        bad = np.nonzero(g[k] != r[k])[0]
        print(v, k, len(bad), [(int(i), g[k][i], r[k][i], g["node"][i], r["node"][i]) for i in bad[:5]])
