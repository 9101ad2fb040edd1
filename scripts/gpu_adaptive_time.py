"""Adaptive mode (main.rs:67-101) at 1080p on the teapot scene: rays/s of hpt_render_device(adaptive) vs the fixed-spp render, both through hpt_render (host image out)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
import torch, hpt
from hpt import scene_io, host_bvh
desc = scene_io.load_scene(os.path.join(ROOT, "tests/golden/scene.xml"))
desc.camera.width, desc.camera.height = 1920, 1080
ctx = hpt.Context(0)
sc = hpt.Scene(ctx, desc, host_bvh=lambda m: host_bvh.build_reference_bvh(m.verts, m.tris))
out = torch.zeros(1920 * 1080 * 3, device="cuda"); torch.cuda.synchronize()
for kw in (dict(spp=256), dict(adaptive=True, initial=16, max_samples=1024), dict(adaptive=True, initial=128, max_samples=512), dict(adaptive=True, initial=16, max_samples=256)):
    sc.render(max_depth=8, seed=1, **kw)
    t0 = time.perf_counter()
    st = sc.render(max_depth=8, seed=1, **kw)["stats"]
    dt = time.perf_counter() - t0
    rays = st.rays_primary + st.rays_bounce
    print(kw, "%.3f s wall, %.3f s stats, %.1f Mrays/s, samples %d (%.1f per px), passes %d" % (dt, st.seconds_total, rays / dt / 1e6, st.samples, st.samples / (1920 * 1080), st.passes), flush=True)
