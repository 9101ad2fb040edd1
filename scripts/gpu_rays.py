"""k_trace_rays rate on the 10 M seeded parity rays (device buffers, HIP-event kernel time), host tree and library tree."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import hpt
from hpt import scene_io, host_bvh
import rays as R

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
desc = scene_io.load_scene(os.path.join(ROOT, "tests/golden/scene.xml"))
ctx = hpt.Context(0)
m = desc.meshes[0]
box = np.concatenate([m.verts.min(0), m.verts.max(0)]).astype(np.float32)
org, dirs = R.parity_rays(box, n)
o4 = torch.zeros((n, 4), dtype=torch.float32); o4[:, :3] = torch.from_numpy(org)
d4 = torch.zeros((n, 4), dtype=torch.float32); d4[:, :3] = torch.from_numpy(dirs)
o4, d4 = o4.cuda(), d4.cuda()
hit = torch.zeros((n, 4), dtype=torch.float32, device="cuda"); side = torch.zeros(n, dtype=torch.uint8, device="cuda")
for name, tree in (("host tree", host_bvh.build_reference_bvh(m.verts, m.tris)), ("SAH tree", None)):
    sc = hpt.Scene(ctx)
    mid = sc.add_mesh(m, tree)
    sc.commit()
    for rep in range(3):
        st = sc.trace_rays_device(mid, n, o4.data_ptr(), d4.data_ptr(), hit.data_ptr(), side.data_ptr(), want_stats=True)
    print("%-10s %8.1f Mrays/s  nodes/ray %.2f tris/ray %.2f  checksum %d" % (name, n / st.seconds_kernel / 1e6, st.nodes_visited / n, st.tris_tested / n,
          int(hit[:, 3].contiguous().view(torch.int32).to(torch.int64).sum().item())), flush=True)
    sc.close()
