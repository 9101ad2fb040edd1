"""Capacity check: a triangle soup of N million unique triangles (default 8) -- commit time, memory, Mrays/s."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, hpt, scenes
n = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 8_000_000
d = scenes.soup(n, w=1920, h=1080)
ctx = hpt.Context(0)
t0 = time.perf_counter(); sc = hpt.Scene(ctx, d); t1 = time.perf_counter() - t0
info = sc.info()
out = torch.zeros(1920 * 1080 * 3, device="cuda")
torch.cuda.synchronize()      # torch fills on its own stream; the library works on the context stream
sc.render_device(out.data_ptr(), want_stats=True, spp=2, max_depth=8, seed=1)
st = sc.render_device(out.data_ptr(), want_stats=True, spp=8, max_depth=8, seed=1, flags=hpt.RENDER_TIME_KERNELS)
rays = st.rays_primary + st.rays_bounce
print("tris %d commit %.2fs scene %.0f MB node_bytes %d stack %d | %.1f Mrays/s (extend %.3fs of %.3fs) image mean %.4f" % (
    n, t1, info.device_bytes / 1e6, info.node_bytes, info.stack_entries, rays / st.seconds_total / 1e6, st.seconds_extend, st.seconds_total,
    float(out.mean()) / 8), flush=True)
