"""Would storing the triangles in tree-leaf order pay?  Emulation: the 1 M-triangle soup with its FACES renumbered along a Morton curve of their centroids
(faces that are neighbours in space -- siblings and cousins in any sensible tree -- then sit in neighbouring 64-B records) against the soup as generated
(random face order): frame time and k_extend launch time, two passes in flight / one stream.   usage: gpu_soup_face_order.py [spp]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, hpt, scenes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
def part1by2(x):
    x = x & 0x3ff
    x = (x | (x << 16)) & 0x030000FF; x = (x | (x << 8)) & 0x0300F00F; x = (x | (x << 4)) & 0x030C30C3; x = (x | (x << 2)) & 0x09249249
    return x
ctx = hpt.Context(0)
for name in ("random face order", "Morton face order", "random face order", "Morton face order"):
    d = scenes.soup(1_000_000, w=1920, h=1080)
    if name.startswith("Morton"):
        m = d.meshes[0]
        c = m.verts[m.tris].mean(axis=1)                      # centroids
        q = np.clip((c / 100.0 * 1024).astype(np.int64), 0, 1023)
        order = np.argsort(part1by2(q[:, 0]) | (part1by2(q[:, 1]) << 1) | (part1by2(q[:, 2]) << 2), kind="stable")
        m.tris = np.ascontiguousarray(m.tris[order]); m.ntris = np.ascontiguousarray(m.ntris[order]); m.ttris = np.ascontiguousarray(m.ttris[order])
    sc = hpt.Scene(ctx, d)
    out = torch.zeros(1920 * 1080 * 3, device="cuda"); torch.cuda.synchronize()
    sc.render_device(out.data_ptr(), want_stats=True, spp=16, max_depth=8, seed=1)
    k = sc.render_device(out.data_ptr(), want_stats=True, spp=spp, max_depth=8, seed=1, flags=hpt.RENDER_TIME_KERNELS)
    t0 = time.perf_counter(); st = sc.render_device(out.data_ptr(), want_stats=True, spp=spp, max_depth=8, seed=1); dt = time.perf_counter() - t0
    print("%-18s frame %8.2f ms %8.1f Mrays/s | one stream: k_extend %.3f ms per launch | image mean %.6f" % (
        name, dt * 1e3, (st.rays_primary + st.rays_bounce) / dt / 1e6, k.seconds_extend / max(k.launches_extend, 1) * 1e3, float(out.mean())), flush=True)
    sc.close()
ctx.close()
