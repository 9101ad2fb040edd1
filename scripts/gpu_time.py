"""Timing experiments: 1080p, 16 spp, kernel times by scene variant."""
import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
import numpy as np, torch
import hpt
from hpt import scene_io, host_bvh

def run(name, desc, tree=True, spp=16, S=4, depth=8, bpc=None):
    ctx = run.ctx
    if bpc is not None: hpt.lib.hpt_debug_set_extend_blocks(ctx.h, bpc)
    sc = hpt.Scene(ctx, desc, host_bvh=(lambda m: host_bvh.build_reference_bvh(m.verts, m.tris)) if tree else None)
    out = torch.zeros(desc.camera.width * desc.camera.height * 3, device="cuda")
    torch.cuda.synchronize()      # torch fills on its own stream; the library works on the context stream
    sc.render_device(out.data_ptr(), want_stats=True, spp=4, max_depth=depth, seed=1, spp_per_pass=S)
    cspp = int(os.environ.get("HPT_TIME_COUNTER_SPP", "4"))      # samples of the instrumented run (its pass size changes the tail share)
    cst = sc.render_device(out.data_ptr(), want_stats=True, spp=cspp, max_depth=depth, seed=1, spp_per_pass=max(S, cspp), flags=hpt.RENDER_COUNTERS)
    st = sc.render_device(out.data_ptr(), want_stats=True, spp=spp, max_depth=depth, seed=1, spp_per_pass=S, flags=hpt.RENDER_TIME_KERNELS)
    rays = st.rays_primary + st.rays_bounce; cr = cst.rays_primary + cst.rays_bounce
    print("%-28s total %.3fs %7.1f Mrays/s | analytic %.4f extend %.4f (%.2f Grays/s) shade %.4f (%.1f ps/ray) | nodes/ray %.2f tris/ray %.2f inst/ray %.1f" % (
        name, st.seconds_total, rays / st.seconds_total / 1e6, st.seconds_analytic, st.seconds_extend, rays / max(st.seconds_extend + st.seconds_analytic, 1e-9) / 1e9, st.seconds_shade, st.seconds_shade / rays * 1e12,
        cst.nodes_visited / cr, cst.tris_tested / cr, cst.inst_tests / cr), flush=True)
    sc.close()

def main():
    run.ctx = hpt.Context(0)
    base = scene_io.load_scene(os.path.join(ROOT, "tests/golden/scene.xml"))
    base.camera.width, base.camera.height = 1920, 1080
    which = sys.argv[1:] or ["full", "sah", "noteapot", "nosphere", "S1", "S8", "S16"]
    for w in which:
        d = copy.deepcopy(base)
        if w == "full": run("full host tree", d)
        elif w == "sah": run("full SAH tree", d, tree=False)
        elif w == "noteapot":
            d.nodes = [n for n in d.nodes if n.geom != 3]; d.meshes = []
            run("no teapot", d)
        elif w == "wallsonly":
            d.nodes = [n for n in d.nodes if n.geom not in (1, 3)]; d.meshes = []
            run("planes only", d)
        elif w == "nosphere":
            d.nodes = [n for n in d.nodes if n.geom != 1]
            run("no sphere", d)
        elif w.startswith("R"):
            for rep in range(2):
                for r in [int(c) for c in w[1:].split(",")]:
                    hpt.lib.hpt_debug_set_refill(r)
                    run("refill %d (rep %d)" % (r, rep), d)
            hpt.lib.hpt_debug_set_refill(24)
        elif w.startswith("NTB"):      # phase 1 alone (no mesh) at several occupancies
            d.nodes = [n for n in d.nodes if n.geom != 3]; d.meshes = []
            for b in [int(c) for c in w[3:].split(",")]:
                run("no teapot, blocks/CU %d" % b, d, bpc=b)
            hpt.lib.hpt_debug_set_extend_blocks(run.ctx.h, 0)
        elif w.startswith("B"):
            for b in [int(c) for c in w[1:].split(",")]:
                run("blocks/CU %d" % b, d, bpc=b)
            hpt.lib.hpt_debug_set_extend_blocks(run.ctx.h, 0)
        elif w.startswith("S"):
            run("full, spp_per_pass=%s" % w[1:], d, S=int(w[1:]))
main()
