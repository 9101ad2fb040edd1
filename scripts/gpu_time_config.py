"""Kernel times of one of the synthetic BASELINE configs at full scene size: gpu_time_config.py <soup|instanced|spheres> [spp] [spp per pass]
(one stream, HIP events per kernel; then the same frame with two passes in flight).  HPT_LIB_PATH selects an A/B build."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import hpt, scenes
from hpt import scene_io, host_bvh
config = sys.argv[1]
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
S = int(sys.argv[3]) if len(sys.argv) > 3 else 0
tree, depth = None, 8
if config == "soup":
    desc = scenes.soup(1_000_000, w=1920, h=1080)
elif config == "instanced":
    teapot = scene_io.load_scene(os.path.join(ROOT, "tests/golden/scene.xml"))
    desc = scenes.instanced(teapot.meshes[0], 15, w=3840, h=2160)
    tree = lambda m: host_bvh.build_reference_bvh(m.verts, m.tris)
else:
    desc, depth = scenes.spheres(10_000, w=1920, h=1080), 16
ctx = hpt.Context(0)
sc = hpt.Scene(ctx, desc, host_bvh=tree)
out = torch.zeros(desc.camera.width * desc.camera.height * 3, device="cuda")
torch.cuda.synchronize()
sc.render_device(out.data_ptr(), want_stats=True, spp=min(spp, 4), max_depth=depth, seed=1, spp_per_pass=S)
cst = sc.render_device(out.data_ptr(), want_stats=True, spp=min(spp, 4), max_depth=depth, seed=1, spp_per_pass=S, flags=hpt.RENDER_COUNTERS)
st = sc.render_device(out.data_ptr(), want_stats=True, spp=spp, max_depth=depth, seed=1, spp_per_pass=S, flags=hpt.RENDER_TIME_KERNELS)
t0 = time.perf_counter()
s2 = sc.render_device(out.data_ptr(), want_stats=True, spp=spp, max_depth=depth, seed=1, spp_per_pass=S)
dt = time.perf_counter() - t0
rays, cr = st.rays_primary + st.rays_bounce, cst.rays_primary + cst.rays_bounce
print("%-10s one stream %.1f Mrays/s (extend %.4f s = %.3f ms per launch, shade %.4f s) | two in flight %.1f Mrays/s | nodes/ray %.2f tris/ray %.2f inst/ray %.2f | image sum %.6e" % (
    config, rays / st.seconds_total / 1e6, st.seconds_extend, st.seconds_extend / max(st.launches_extend, 1) * 1e3, st.seconds_shade, rays / dt / 1e6,
    cst.nodes_visited / cr, cst.tris_tested / cr, cst.inst_tests / cr, float(out.double().sum())), flush=True)
sc.close()
