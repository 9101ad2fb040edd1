"""Refill threshold of the persistent traversal kernels (idle lanes at which a wave leaves its box-step loop to pull new rays) on a config at full size:
gpu_refill_sweep.py <soup|instanced|spheres|teapot> <spp> [thresholds...]   -- one-stream kernel times (HPT_RENDER_TIME_KERNELS) and the frame with two passes in flight"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, hpt, scenes
from hpt import scene_io, host_bvh
cfg, spp = sys.argv[1], int(sys.argv[2])
ths = [int(x) for x in sys.argv[3:]] or [8, 16, 24, 32, 40]
base = scene_io.load_scene(os.path.join(ROOT, "tests/golden/scene.xml"))
tree = lambda m: host_bvh.build_reference_bvh(m.verts, m.tris)
depth = 8
if cfg == "soup": desc, tree = scenes.soup(1_000_000, w=1920, h=1080), None
elif cfg == "instanced": desc = scenes.instanced(base.meshes[0], 15, w=3840, h=2160)
elif cfg == "spheres": desc, tree, depth = scenes.spheres(10_000, w=1920, h=1080), None, 16
else: desc = base; desc.camera.width, desc.camera.height = 1920, 1080
ctx = hpt.Context(0)
sc = hpt.Scene(ctx, desc, host_bvh=tree)
npix = desc.camera.width * desc.camera.height
out = torch.zeros(npix * 3, device="cuda"); torch.cuda.synchronize()
sc.render_device(out.data_ptr(), want_stats=True, spp=min(spp, 16), max_depth=depth, seed=1)
for rep in range(2):
    for th in ths:
        hpt.lib.hpt_debug_set_refill(th)
        k = sc.render_device(out.data_ptr(), want_stats=True, spp=spp, max_depth=depth, seed=1, flags=hpt.RENDER_TIME_KERNELS)
        t0 = time.perf_counter()
        st = sc.render_device(out.data_ptr(), want_stats=True, spp=spp, max_depth=depth, seed=1)
        dt = time.perf_counter() - t0
        print("%s refill %2d (rep %d): frame %8.2f ms %8.1f Mrays/s | one stream: extend %.2f ms per launch" % (
            cfg, th, rep, dt * 1e3, (st.rays_primary + st.rays_bounce) / dt / 1e6, k.seconds_extend / max(k.launches_extend, 1) * 1e3), flush=True)
hpt.lib.hpt_debug_set_refill(24)
sc.close(); ctx.close()
