"""Upper bound for sorting the rays of a bounce depth: the 1 M-triangle soup, 16 M bounce-like rays (origins uniform in the
soup's cube, directions uniform on the sphere -- what a diffuse bounce off a random triangle looks like) through the Mesh::intersect entry
(hpt_trace_rays_device: root-box pass + the persistent traversal kernel), traced
   (a) in random order (what the wavefront queues hold after a bounce),
   (b) sorted by direction octant,
   (c) sorted by origin cell (Morton code of a G^3 grid) and octant, G = 2 ... 64 (4^3 = the 64 segments a queue has),
the sort itself NOT timed (torch.argsort on the device): the kernel-time ratio is what a free sort would buy.  Hit records are compared
between the orders (same rays -> same hits).    usage: gpu_sort_ab.py [n_rays] [n_tris]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import hpt, scenes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16_000_000
ntri = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
ctx = hpt.Context(0)
g = torch.Generator(device="cuda"); g.manual_seed(7)
if ntri > 0:
    desc = scenes.soup(ntri, w=1920, h=1080)
    E = 100.0 * (ntri / 1.0e6) ** (1.0 / 3.0)
    sc = hpt.Scene(ctx, desc, host_bvh=None)
    o = torch.rand((n, 3), generator=g, device="cuda") * E
    d = torch.randn((n, 3), generator=g, device="cuda"); d = d / d.norm(dim=1, keepdim=True)
    lo = torch.zeros(3, device="cuda")
else:      # n_tris = 0: the teapot (L2-resident tree), the parity rays of SURVEY 8d.6 (tests/rays.py) in shuffled order
    from hpt import scene_io, host_bvh
    import rays as R
    desc = scene_io.load_scene(os.path.join(ROOT, "tests/golden/scene.xml"))
    sc = hpt.Scene(ctx, desc, host_bvh=lambda m: host_bvh.build_reference_bvh(m.verts, m.tris))
    v = desc.meshes[0].verts
    box = (v.min(axis=0), v.max(axis=0))
    org, dirs = R.parity_rays(np.concatenate(box), n)
    perm = torch.randperm(n, generator=g, device="cuda")
    o = torch.from_numpy(org).cuda()[perm]; d = torch.from_numpy(dirs).cuda()[perm]
    c0 = torch.from_numpy((box[0] + box[1]) / 2).cuda(); half = float(2.6 * np.linalg.norm(box[1] - box[0]) / 2)
    lo = c0 - half; E = 2 * half      # cube around the sphere the outside origins sit on


def part1by2(x):
    x = x & 0x3ff
    x = (x | (x << 16)) & 0x030000FF
    x = (x | (x << 8)) & 0x0300F00F
    x = (x | (x << 4)) & 0x030C30C3
    x = (x | (x << 2)) & 0x09249249
    return x


def trace(order, name):
    oo = torch.zeros((n, 4), device="cuda"); dd = torch.zeros((n, 4), device="cuda")
    oo[:, :3] = o[order] if order is not None else o
    dd[:, :3] = d[order] if order is not None else d
    hit = torch.zeros((n, 4), device="cuda"); side = torch.zeros(n, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    cst = sc.trace_rays_device(0, n, oo.data_ptr(), dd.data_ptr(), hit.data_ptr(), side.data_ptr(), count=True, want_stats=True)
    best = None
    for _ in range(3):
        st = sc.trace_rays_device(0, n, oo.data_ptr(), dd.data_ptr(), hit.data_ptr(), side.data_ptr(), want_stats=True)
        best = st.seconds_kernel if best is None else min(best, st.seconds_kernel)
    torch.cuda.synchronize()
    print("%-34s kernel %8.3f ms  %8.1f Mrays/s | nodes/ray %.2f tris/ray %.2f" % (name, best * 1e3, n / best / 1e6, cst.nodes_visited / n, cst.tris_tested / n), flush=True)
    back = hit
    if order is not None:
        back = torch.empty_like(hit); back[order] = hit
    return best, back


t0, h0 = trace(None, "random order")
octant = ((d[:, 0] < 0).long() | ((d[:, 1] < 0).long() << 1) | ((d[:, 2] < 0).long() << 2))
t1, h1 = trace(torch.argsort(octant, stable=True), "sorted by octant")
assert torch.equal(h0.view(torch.int32), h1.view(torch.int32))
for G in (2, 4, 8, 16, 32, 64):
    c = ((o - lo) / E * G).long().clamp_(0, G - 1)
    morton = part1by2(c[:, 0]) | (part1by2(c[:, 1]) << 1) | (part1by2(c[:, 2]) << 2)
    for with_oct in (False, True):
        key = (morton << 3) | octant if with_oct else morton
        t, h = trace(torch.argsort(key, stable=True), "sorted by cell %d^3%s" % (G, " + octant" if with_oct else ""))
        assert torch.equal(h0.view(torch.int32), h.view(torch.int32))
        print("    -> %.2fx the random order" % (t0 / t), flush=True)
sc.close(); ctx.close()
