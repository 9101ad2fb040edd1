"""First-contact GPU script: parity of trace / scene-intersect / render against the oracle + a timing run."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hpt
from hpt import scene_io, host_bvh
import oracle_lib as ol
import rays as R

def main():
    n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    desc = scene_io.load_scene(os.path.join(ROOT, "tests/golden/scene.xml"))
    olib = ol.load()
    osc = ol.OracleScene(olib, desc)
    ctx = hpt.Context(0)
    sc = hpt.Scene(ctx, desc, host_bvh=lambda m: host_bvh.build_reference_bvh(m.verts, m.tris))
    print("scene committed", flush=True)
    # 1. mesh parity
    org, dirs = R.parity_rays(osc.meshes[0].box(), n_rays)
    t0 = time.time(); g = sc.trace_rays(0, org, dirs); t1 = time.time()
    o = osc.meshes[0].intersect(org, dirs); t2 = time.time()
    st = g["stats"]
    print("trace: gpu %.3fs (kernel %.4fs, %.1f Mrays/s) oracle %.2fs (%d thr)" % (t1 - t0, st.seconds_kernel, n_rays / st.seconds_kernel / 1e6, t2 - t1, o["stats"].threads))
    print("  gpu nodes/ray %.1f tris/ray %.2f | oracle pops/ray %.1f tris/ray %.2f" % (st.nodes_visited / n_rays, st.tris_tested / n_rays, o["stats"].node_pops / n_rays, o["stats"].tri_tests / n_rays))
    same = g["prim"] == o["prim"]
    hit = o["prim"] >= 0
    print("  hits %.3f  id mismatches %d" % (hit.mean(), (~same).sum()))
    bad = ~same
    if bad.any():
        ed = R.edge_distance(o["u"][bad], o["v"][bad]); ed2 = R.edge_distance(g["u"][bad], g["v"][bad])
        print("   mismatch: oracle miss %d gpu miss %d; min edge dist (oracle) max %.3g ; t rel diff max %.3g" % ((o["prim"][bad] < 0).sum(), (g["prim"][bad] < 0).sum(), np.minimum(ed, ed2).max(), np.abs(g["t"][bad] - o["t"][bad]).max()))
        for k in np.nonzero(bad)[0][:10]:
            print("    ray", k, "gpu", g["prim"][k], g["t"][k], g["u"][k], g["v"][k], "orc", o["prim"][k], o["t"][k], o["u"][k], o["v"][k])
    both = same & hit
    print("  bit-exact t %d/%d u %d v %d side %d" % ((g["t"][both] == o["t"][both]).sum(), both.sum(), (g["u"][both] == o["u"][both]).sum(), (g["v"][both] == o["v"][both]).sum(), (g["side"][both] == o["side"][both]).sum()))
    # 2. scene intersect parity: primary rays of a coarse grid + bounce-like random rays
    rng = np.random.default_rng(7)
    n2 = 200000
    so = np.stack([rng.uniform(-14, 14, n2), rng.uniform(-58, 19, n2), rng.uniform(0.5, 23, n2)], 1).astype(np.float32)
    sd = rng.normal(size=(n2, 3)); sd = (sd / np.linalg.norm(sd, axis=1, keepdims=True)).astype(np.float32)
    gi = sc.intersect(so, sd); oi = osc.intersect(so, sd)
    for k in ("node", "prim", "t", "side", "pos", "normal", "uvw"):
        eq = (gi[k] == oi[k]); eq = eq.all(axis=1) if eq.ndim == 2 else eq
        md = np.abs(gi[k].astype(np.float64) - oi[k].astype(np.float64)).max()
        print("  scene_intersect %-6s equal %d/%d maxdiff %.3g" % (k, eq.sum(), n2, md))
    # 3. render parity, same RNG
    import copy
    small = copy.deepcopy(desc); small.camera.width, small.camera.height = 160, 120
    osm = ol.OracleScene(olib, small)
    ssm = hpt.Scene(ctx, small, host_bvh=lambda m: host_bvh.build_reference_bvh(m.verts, m.tris))
    spp = 16
    t0 = time.time(); orr = osm.render(spp=spp, max_depth=8, seed=1, want_samples=True); t1 = time.time()
    gs = ssm.render_samples(spp=spp, max_depth=8, seed=1)
    gr = ssm.render(spp=spp, max_depth=8, seed=1, flags=hpt.RENDER_COUNTERS)
    ops = orr["per_sample"]
    rel = np.abs(gs - ops) / (np.abs(ops) + 1e-3)
    print("render 160x120x%d: oracle %.2fs rays %d | gpu rays %d" % (spp, t1 - t0, orr["stats"].rays, gr["stats"].rays_primary + gr["stats"].rays_bounce))
    print("  per-sample: exact %d / %d, rel>1e-3: %d, max rel %.3g" % ((gs == ops).all(axis=1).sum() if False else (gs == ops).all(axis=2).sum(), gs.shape[0] * gs.shape[1], (rel.max(axis=2) > 1e-3).sum(), rel.max()))
    d = gr["rgb"] - orr["rgb"]
    print("  image: mean gpu %s oracle %s rmse %.4g rel-rmse %.4g" % (gr["rgb"].mean(axis=(0, 1)), orr["rgb"].mean(axis=(0, 1)), np.sqrt((d ** 2).mean()), np.sqrt((d ** 2).mean()) / orr["rgb"].mean()))
    print("  gpu nodes/ray %.1f tris/ray %.2f inst/ray %.2f" % (gr["stats"].nodes_visited / (gr["stats"].rays_primary + gr["stats"].rays_bounce), gr["stats"].tris_tested / (gr["stats"].rays_primary + gr["stats"].rays_bounce), gr["stats"].inst_tests / (gr["stats"].rays_primary + gr["stats"].rays_bounce)))
    # 4. timing at 1080p
    big = copy.deepcopy(desc); big.camera.width, big.camera.height = 1920, 1080
    sb = hpt.Scene(ctx, big, host_bvh=lambda m: host_bvh.build_reference_bvh(m.verts, m.tris))
    import torch
    out = torch.zeros(1920 * 1080 * 3, device="cuda", dtype=torch.float32)
    torch.cuda.synchronize()      # torch fills on its own stream; the library works on the context stream
    for spp_pp in (1, 4, 8):
        sb.render_device(out.data_ptr(), want_stats=True, spp=8, max_depth=8, seed=1, spp_per_pass=spp_pp)
        st = sb.render_device(out.data_ptr(), want_stats=True, spp=16, max_depth=8, seed=1, spp_per_pass=spp_pp, flags=hpt.RENDER_TIME_KERNELS)
        rays = st.rays_primary + st.rays_bounce
        print("1080p 16spp (S=%d): total %.3fs rays %d -> %.1f Mrays/s | raygen %.3f extend %.3f shade %.3f resolve %.3f (launches %d)" % (
            spp_pp, st.seconds_total, rays, rays / st.seconds_total / 1e6, st.seconds_raygen, st.seconds_extend, st.seconds_shade, st.seconds_resolve, st.launches_extend), flush=True)
    print("mean", out.view(-1, 3).mean(dim=0).cpu().numpy() / 16)

main()
