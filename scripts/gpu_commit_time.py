"""Host time of hpt.Scene(...) (add meshes + hpt_scene_commit: validation, tree restructuring, flatten, upload) per config."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hpt, scenes
from hpt import scene_io, host_bvh
ctx = hpt.Context(0)
teapot = scene_io.load_scene(os.path.join(ROOT, "tests/golden/scene.xml"))
for name, make, tree in (("teapot (host bvh.rs tree handed over)", lambda: teapot, True), ("soup 1 M triangles (library SAH)", lambda: scenes.soup(1_000_000, w=64, h=64), False),
                         ("instanced 3 375 teapots", lambda: scenes.instanced(teapot.meshes[0], 15, w=64, h=64), True), ("spheres 10 k", lambda: scenes.spheres(10_000, w=64, h=64), False)):
    d = make()
    t0 = time.perf_counter()
    hb = (lambda m: host_bvh.build_reference_bvh(m.verts, m.tris)) if tree else None
    sc = hpt.Scene(ctx, d, host_bvh=hb)
    dt = time.perf_counter() - t0
    i = sc.info()
    print("%-40s %.2f s  (%d triangles, %d nodes, %d instances)" % (name, dt, i.n_triangles, i.n_nodes, i.n_instances), flush=True)
    sc.close()
