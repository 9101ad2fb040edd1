"""Short workload for rocprofv3 counter passes: gpu_prof.py <spp> <spp per pass> [config]
   teapot (default): 1080p, e.g. one pass of 4 spp (9 k_extend + 9 k_shade launches); soup | instanced | spheres: the BASELINE
   configs at full scene and frame size (tests/scenes.py generators, library-built trees)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import hpt
from hpt import scene_io, host_bvh
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 4
S = int(sys.argv[2]) if len(sys.argv) > 2 else 4      # samples per pass (4 = 8 M paths; 32 = the bench's 66 M)
config = sys.argv[3] if len(sys.argv) > 3 else "teapot"
desc = scene_io.load_scene(os.path.join(ROOT, "tests/golden/scene.xml"))
tree = lambda m: host_bvh.build_reference_bvh(m.verts, m.tris)
depth = 8
if config == "teapot":
    desc.camera.width, desc.camera.height = 1920, 1080
else:
    import scenes
    if config == "soup":
        desc, tree = scenes.soup(1_000_000, w=1920, h=1080), None
    elif config == "instanced":
        desc = scenes.instanced(desc.meshes[0], 15, w=3840, h=2160)
    else:
        desc, tree, depth = scenes.spheres(10_000, w=1920, h=1080), None, 16
ctx = hpt.Context(0)
sc = hpt.Scene(ctx, desc, host_bvh=tree)
out = torch.zeros(desc.camera.width * desc.camera.height * 3, device="cuda")
torch.cuda.synchronize()      # torch fills on its own stream; the library works on the context stream
st = sc.render_device(out.data_ptr(), want_stats=True, spp=spp, max_depth=depth, seed=1, spp_per_pass=S)
print("rays", st.rays_primary + st.rays_bounce, "seconds", st.seconds_total)
sc.close(); ctx.close()
