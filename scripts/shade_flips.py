import os, sys
sys.path.insert(0, "path-tracer_amd"); sys.path.insert(0, "tests")
import numpy as np, hpt
from hpt import scene_io, host_bvh
import oracle_lib as ol
desc = scene_io.load_scene("tests/golden/scene.xml"); desc.camera.width, desc.camera.height = 192, 144
ctx = hpt.Context(0)
sc = hpt.Scene(ctx, desc, host_bvh=lambda m: host_bvh.build_reference_bvh(m.verts, m.tris))
osc = ol.OracleScene(ol.load(), desc)
spp = 8
ref = osc.render(spp=spp, max_depth=8, seed=1, want_samples=True, threads=16)
for name, flags in (("exact", hpt.RENDER_EXACT_SHADE), ("default", 0)):
    got = sc.render_samples(spp=spp, max_depth=8, seed=1, flags=flags)
    rel = np.abs(got - ref["per_sample"]).max(axis=2) / (np.abs(ref["per_sample"]).max(axis=2) + 1e-3)
    st = sc.render(spp=spp, max_depth=8, seed=1, flags=flags)
    img = st["rgb"]; st = st["stats"]
    rm = float(np.sqrt(((img - ref["rgb"]) ** 2).mean()) / ref["rgb"].mean())
    print("%-8s samples off by > 1e-3: %d of %d (%.2e), > 1e-5: %.2e; rays %d vs %d; image rel rmse %.2e" % (name, (rel > 1e-3).sum(), rel.size, (rel > 1e-3).mean(), (rel > 1e-5).mean(), st.rays_primary + st.rays_bounce, ref["stats"].rays, rm))
