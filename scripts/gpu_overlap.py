"""A/B of several passes in flight (hpt_debug_set_overlap) on the bench frame: wall time per frame, images compared bit for bit."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
import torch
import hpt
from hpt import scene_io, host_bvh

def main():
    spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    spps = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
    tree = (sys.argv[3] if len(sys.argv) > 3 else "host") == "host"
    # modes: in-flight passes + 10 * extend blocks per CU (0 = default); 0 = one stream
    modes = [int(v) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 else [0, 2]
    ctx = hpt.Context(0)
    stream = torch.cuda.Stream()
    ctx.set_stream(stream.cuda_stream)
    d = scene_io.load_scene(os.path.join(ROOT, "tests/golden/scene.xml"))
    d.camera.width, d.camera.height = 1920, 1080
    sc = hpt.Scene(ctx, d, host_bvh=(lambda m: host_bvh.build_reference_bvh(m.verts, m.tris)) if tree else None)
    with torch.cuda.stream(stream):
        outs = [torch.zeros(1920 * 1080 * 3, device="cuda") for _ in range(2)]
    if os.environ.get("HPT_REFILL"): hpt.lib.hpt_debug_set_refill(int(os.environ["HPT_REFILL"]))
    sc.render_device(outs[0].data_ptr(), want_stats=True, spp=64, max_depth=8, seed=1)
    for S in spps:
        for rep in range(2):
            for on in modes:
                hpt.lib.hpt_debug_set_overlap(ctx.h, on % 10 if on else 1)
                hpt.lib.hpt_debug_set_extend_blocks(ctx.h, on // 10)
                stream.synchronize(); t0 = time.perf_counter()
                st = sc.render_device(outs[1 if on else 0].data_ptr(), want_stats=True, spp=spp, max_depth=8, seed=1, spp_per_pass=S)
                stream.synchronize(); dt = time.perf_counter() - t0
                rays = st.rays_primary + st.rays_bounce
                print("spp_per_pass %2d in flight %d (blocks/CU %d): %.4f s  %8.1f Mrays/s  passes %d" % (S, on % 10, on // 10, dt, rays / dt / 1e6, st.passes), flush=True)
            print("   images identical:", bool(torch.equal(outs[0], outs[1])), flush=True)
    sc.close()
main()
