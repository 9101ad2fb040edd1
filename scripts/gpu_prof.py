"""Short workload for rocprofv3 counter passes: one 1080p pass of 4 spp (9 k_extend + 9 k_shade launches)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
import torch
import hpt
from hpt import scene_io, host_bvh
desc = scene_io.load_scene(os.path.join(ROOT, "tests/golden/scene.xml"))
desc.camera.width, desc.camera.height = 1920, 1080
ctx = hpt.Context(0)
sc = hpt.Scene(ctx, desc, host_bvh=lambda m: host_bvh.build_reference_bvh(m.verts, m.tris))
out = torch.zeros(1920 * 1080 * 3, device="cuda")
torch.cuda.synchronize()      # torch fills on its own stream; the library works on the context stream
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 4
S = int(sys.argv[2]) if len(sys.argv) > 2 else 4      # samples per pass (4 = 8 M paths; 32 = the bench's 66 M)
st = sc.render_device(out.data_ptr(), want_stats=True, spp=spp, max_depth=8, seed=1, spp_per_pass=S)
print("rays", st.rays_primary + st.rays_bounce, "seconds", st.seconds_total)
sc.close(); ctx.close()
