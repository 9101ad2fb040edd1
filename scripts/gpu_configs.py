"""BASELINE.json configs 3-5 at full size on one MI355X (bounded spp): commit time, Mrays/s, kernel split, counters.
  python3 scripts/gpu_configs.py [soup] [inst] [spheres] [teapot]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import hpt
from hpt import scene_io, host_bvh
import scenes


def run(ctx, name, desc, spp, depth, tree=None, S=0):
    t0 = time.perf_counter()
    sc = hpt.Scene(ctx, desc, host_bvh=tree)
    t_commit = time.perf_counter() - t0
    npix = desc.camera.width * desc.camera.height
    out = torch.zeros(npix * 3, device="cuda")
    torch.cuda.synchronize()      # torch fills on its own stream; the library works on the context stream
    sc.render_device(out.data_ptr(), want_stats=True, spp=min(spp, 4), max_depth=depth, seed=1, spp_per_pass=S)
    cst = sc.render_device(out.data_ptr(), want_stats=True, spp=min(spp, 4), max_depth=depth, seed=1, spp_per_pass=S, flags=hpt.RENDER_COUNTERS)
    st = sc.render_device(out.data_ptr(), want_stats=True, spp=spp, max_depth=depth, seed=1, spp_per_pass=S, flags=hpt.RENDER_TIME_KERNELS)
    rays = st.rays_primary + st.rays_bounce; cr = cst.rays_primary + cst.rays_bounce
    # library default (two passes in flight; needs >= 2 passes): wall clock of a render without the event timing
    torch.cuda.synchronize(); t1 = time.perf_counter()
    wst = sc.render_device(out.data_ptr(), want_stats=True, spp=2 * spp, max_depth=depth, seed=1, spp_per_pass=S)
    torch.cuda.synchronize(); wall = time.perf_counter() - t1
    info = sc.info()
    bpr = (info.node_bytes * float(cst.nodes_visited) + 48.0 * cst.tris_tested + (128.0 * cst.inst_tests if info.uses_tlas else 0.0)) / cr + 52
    rec = {"config": name, "res": [desc.camera.width, desc.camera.height], "spp": spp, "depth": depth, "commit_s": round(t_commit, 3), "scene_MB": round(info.device_bytes / 1e6, 1), "tlas": int(info.uses_tlas),
           "seconds": round(st.seconds_total, 4), "Mrays_s": round((wst.rays_primary + wst.rays_bounce) / wall / 1e6, 1), "passes": int(wst.passes),
           "Mrays_s_one_stream": round(rays / st.seconds_total / 1e6, 1), "rays_per_sample": round(rays / (npix * spp), 3),
           "extend_s": round(st.seconds_extend, 4), "shade_s": round(st.seconds_shade, 4), "extend_Grays_s": round(rays / st.seconds_extend / 1e9, 3),
           "nodes_per_ray": round(cst.nodes_visited / cr, 2), "tris_per_ray": round(cst.tris_tested / cr, 2), "inst_per_ray": round(cst.inst_tests / cr, 2),
           "bytes_per_ray": round(bpr, 1), "extend_GBs": round(bpr * rays / st.seconds_extend / 1e9, 1),
           "image_mean": [round(float(v), 5) for v in (out.view(-1, 3).mean(dim=0) / (2 * spp)).cpu()]}
    print(json.dumps(rec), flush=True)
    sc.close()


def main():
    ctx = hpt.Context(0)
    which = sys.argv[1:] or ["teapot", "soup", "inst", "spheres"]
    base = scene_io.load_scene(os.path.join(ROOT, "tests/golden/scene.xml"))
    ref_tree = lambda m: host_bvh.build_reference_bvh(m.verts, m.tris)
    for w in which:
        if w == "teapot":
            base.camera.width, base.camera.height = 1920, 1080
            run(ctx, "2 teapot 1080p depth 8 (host tree)", base, 64, 8, tree=ref_tree)
            run(ctx, "2 teapot 1080p depth 8 (SAH)", base, 64, 8)
        elif w == "soup":
            t0 = time.perf_counter(); d = scenes.soup(1000000, w=1920, h=1080); print("soup gen %.1fs" % (time.perf_counter() - t0), flush=True)
            run(ctx, "3 soup 1M tris 1080p depth 8 (SAH)", d, 32, 8)
        elif w == "inst":
            d = scenes.instanced(base.meshes[0], 15, w=3840, h=2160)
            n = len(d.nodes) - 1
            run(ctx, "4 instanced %d teapots (%.1fM tris) 4K depth 8 (TLAS+SAH BLAS)" % (n, n * len(base.meshes[0].tris) / 1e6), d, 16, 8)
        elif w == "spheres":
            d = scenes.spheres(10000, w=1920, h=1080)
            run(ctx, "5 10k spheres 1080p depth 16 (TLAS)", d, 32, 16)


main()
