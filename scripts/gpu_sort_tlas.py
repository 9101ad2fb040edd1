"""Ray order on the two-level path (instanced teapots / 10 k spheres): the same N bounce-like rays (origins uniform in the scene's box, directions
uniform) through hpt_scene_intersect in random order and sorted by origin cell (+ octant); run under rocprofv3 --kernel-trace, the k_scene_closest
durations (stream order = call order) are the answer (scripts/gpu_sort_tlas.sh prints them).   usage: gpu_sort_tlas.py <instanced|spheres> [n]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import hpt, scenes
from hpt import scene_io, host_bvh
cfg = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_000
rng = np.random.default_rng(11)
if cfg == "instanced":
    base = scene_io.load_scene(os.path.join(ROOT, "tests/golden/scene.xml"))
    desc = scenes.instanced(base.meshes[0], 15, w=96, h=54)
    lo, hi = np.array([-20, -20, -5], np.float32), np.array([600, 600, 600], np.float32)
    tree = lambda m: host_bvh.build_reference_bvh(m.verts, m.tris)
else:
    desc = scenes.spheres(10_000, w=96, h=54)
    lo, hi = np.array([-50, -50, 0], np.float32), np.array([50, 50, 31], np.float32)
    tree = None
ctx = hpt.Context(0)
sc = hpt.Scene(ctx, desc, host_bvh=tree)
o = (lo + (hi - lo) * rng.random((n, 3), dtype=np.float32)).astype(np.float32)
d = rng.normal(size=(n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
octant = (d[:, 0] < 0).astype(np.int64) | ((d[:, 1] < 0).astype(np.int64) << 1) | ((d[:, 2] < 0).astype(np.int64) << 2)
def part1by2(x):
    x = x & 0x3ff
    x = (x | (x << 16)) & 0x030000FF; x = (x | (x << 8)) & 0x0300F00F; x = (x | (x << 4)) & 0x030C30C3; x = (x | (x << 2)) & 0x09249249
    return x
ref = sc.intersect(o, d)      # warm-up + reference
orders = [("random", None)]
for G in (4, 16, 64):
    c = np.clip(((o - lo) / (hi - lo) * G).astype(np.int64), 0, G - 1)
    m = part1by2(c[:, 0]) | (part1by2(c[:, 1]) << 1) | (part1by2(c[:, 2]) << 2)
    orders.append(("cell %d^3" % G, np.argsort(m, kind="stable")))
    orders.append(("cell %d^3 + octant" % G, np.argsort((m << 3) | octant, kind="stable")))
for name, order in orders:
    r = sc.intersect(o[order], d[order]) if order is not None else sc.intersect(o, d)
    if order is not None:
        assert np.array_equal(r["t"], ref["t"][order]) and np.array_equal(r["node"], ref["node"][order])
    print("order:", name, flush=True)
sc.close(); ctx.close()
