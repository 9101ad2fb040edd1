"""GPU parity of the traversal kernel (k_trace_rays, through the C ABI) against the oracle's Mesh::intersect.
north_star check 1: hit primitive id identical except rays within 1e-6 of an edge; t within 1e-4 relative.
In fact the kernel keeps the reference's operation order for the triangle test, so t,u,v are bit-identical."""
import os

import numpy as np
import pytest

import rays as R
from conftest import GOLDEN, host_tree

pytestmark = pytest.mark.gpu


def compare(g, o):
    same = g["prim"] == o["prim"]
    bad = ~same
    if bad.any():
        # only near-edge hits may pick a different (equal-t) triangle: tie winners depend on visiting order
        ed = np.minimum(R.edge_distance(o["u"][bad], o["v"][bad]), R.edge_distance(g["u"][bad], g["v"][bad]))
        assert (ed < 1e-6).all(), "id mismatch away from an edge: %d rays" % (ed >= 1e-6).sum()
        assert ((g["prim"][bad] >= 0) & (o["prim"][bad] >= 0)).all()
    hit = (o["prim"] >= 0) & (g["prim"] >= 0)
    rel = np.abs(g["t"][hit] - o["t"][hit]) / np.maximum(np.abs(o["t"][hit]), 1e-30)
    assert (rel <= 1e-4).all()                                  # tolerance stated by north_star
    both = same & hit
    for k in ("t", "u", "v", "side"):
        np.testing.assert_array_equal(g[k][both], o[k][both])   # ... and actually bit-exact
    return int(bad.sum())


def test_golden_hits(gscene, oscene):
    g0 = np.load(os.path.join(GOLDEN, "hits_4096.npz"))
    org, dirs = R.parity_rays(oscene.meshes[0].box(), 4096)
    g = gscene.trace_rays(0, org, dirs)
    assert compare(g, {k: g0[k] for k in g0.files}) == 0


def test_one_million_rays(gscene, oscene):
    org, dirs = R.parity_rays(oscene.meshes[0].box(), 1_000_000)
    g = gscene.trace_rays(0, org, dirs)
    o = oscene.meshes[0].intersect(org, dirs)
    compare(g, o)
    st = g["stats"]
    assert st.rays == 1_000_000 and 0 < st.nodes_visited < o["stats"].node_pops    # pruning + tight boxes visit fewer nodes
    assert st.tris_tested <= o["stats"].tri_tests


def test_ten_million_rays_full_size(gscene, oscene):
    """BASELINE size (10 M rays); the oracle needs ~3 s on 16 cores."""
    org, dirs = R.parity_rays(oscene.meshes[0].box(), 10_000_000, seed=55)
    g = gscene.trace_rays(0, org, dirs, want_stats=False)
    o = oscene.meshes[0].intersect(org, dirs)
    compare(g, o)


def test_library_built_tree_gives_same_hits(ctx, desc, oscene):
    """hits do not depend on the tree: nodes=NULL lets the library build its own SAH tree"""
    import hpt
    sc = hpt.Scene(ctx)
    mid = sc.add_mesh(desc.meshes[0], None)
    sc.commit()
    org, dirs = R.parity_rays(oscene.meshes[0].box(), 200_000, seed=3)
    g = sc.trace_rays(mid, org, dirs)
    o = oscene.meshes[0].intersect(org, dirs)
    compare(g, o)
    sc.close()


def test_tree_optimizer_changes_no_result(ctx, desc, oscene):
    """hpt_scene_commit restructures the handed-over bvh.rs tree (SAH re-partition + insertion rounds, csrc/hpt_bvh.cpp
    tree_optimize).  1 M parity rays and a 64x48x16 image must be bit-identical with and without it; it must cut the box
    steps per ray, and stay inside the traversal-stack budget."""
    import hpt
    from conftest import resized
    org, dirs = R.parity_rays(oscene.meshes[0].box(), 1_000_000, seed=77)
    out = {}
    try:
        for mode in (0, 1):
            hpt.set_tree_opt(mode)
            sc = hpt.Scene(ctx, resized(desc, 64, 48), host_bvh=host_tree)
            info = sc.info()
            assert info.trees_optimized == mode and info.stack_entries <= 20
            out[mode] = (sc.trace_rays(0, org, dirs), sc.render(spp=16, max_depth=8, seed=3), sc.render_samples(spp=2, max_depth=8, seed=5))
            sc.close()
    finally:
        hpt.set_tree_opt(-1)
    for k in ("prim", "t", "u", "v", "side"):
        np.testing.assert_array_equal(out[0][0][k], out[1][0][k])
    np.testing.assert_array_equal(out[0][1]["rgb"], out[1][1]["rgb"])
    np.testing.assert_array_equal(out[0][2], out[1][2])
    assert out[0][1]["stats"].rays_bounce == out[1][1]["stats"].rays_bounce
    assert out[1][0]["stats"].nodes_visited < 0.6 * out[0][0]["stats"].nodes_visited
    compare(out[1][0], oscene.meshes[0].intersect(org, dirs))


def test_far_from_the_origin(ctx, olib, desc, monkeypatch):
    """Box culling stays conservative when |origin| >> |t|: the teapot moved far away, rays starting next to it --
    b = fl(o/d) then carries an absolute error far above any relative slack on t (the kernels before the per-axis bound
    lost 7 of 400 k hits here).  Host tree and library tree; Node32 at the distance where its grid still resolves
    (at 5e4 one quantum is below the vertex ulp and commit keeps Node64)."""
    import copy
    import hpt
    import oracle_lib as ol
    for offset, formats in (((3.0e4, -2.0e4, 5.0e4), ("64",)), ((600.0, -400.0, 500.0), ("64", "32"))):
        m = copy.deepcopy(desc.meshes[0])
        m.verts = (m.verts + np.array(offset, np.float32)).astype(np.float32)
        om = ol.OracleMesh(olib, m)
        org, dirs = R.parity_rays(om.box(), 400_000, seed=21)
        o = om.intersect(org, dirs)
        assert (o["prim"] >= 0).mean() > 0.2
        for fmt in formats:
            monkeypatch.setenv("HPT_NODE_FORMAT", fmt)
            for tree in (host_tree(m), None):
                sc = hpt.Scene(ctx)
                mid = sc.add_mesh(m, tree)
                sc.commit()
                assert sc.info().node_bytes == int(fmt)
                compare(sc.trace_rays(mid, org, dirs), o)
                sc.close()


def test_edge_cases(ctx, olib):
    import hpt
    import oracle_lib as ol
    from hpt import scene_io
    f32 = np.float32
    # one-triangle mesh (BVH is a single leaf) and a two-triangle quad sharing an edge
    for tris in ([[0, 1, 2]], [[0, 1, 2], [0, 2, 3]]):
        m = scene_io.Mesh(verts=np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], f32), tris=np.array(tris, np.uint32),
                          normals=np.array([[0, 0, 1]], f32), ntris=np.zeros((len(tris), 3), np.uint32),
                          uvw=np.array([[0, 0, 0]], f32), ttris=np.zeros((len(tris), 3), np.uint32))
        sc = hpt.Scene(ctx)
        mid = sc.add_mesh(m, host_tree(m))
        sc.commit()
        om = ol.OracleMesh(olib, m)
        rng = np.random.default_rng(4)
        org = np.concatenate([rng.uniform(-0.5, 1.5, (5000, 2)), np.full((5000, 1), 2.0)], 1).astype(f32)
        dirs = np.tile(np.array([[0, 0, -1]], f32), (5000, 1))
        org[0] = [0.5, 0.5, 2]; org[1] = [0.25, 0.25, 0.0]; org[2] = [0.25, 0.25, -1]   # diagonal edge, t == 0, behind
        g = sc.trace_rays(mid, org, dirs); o = om.intersect(org, dirs)
        compare(g, o)
        assert g["prim"][2] == -1 and g["t"][1] == 0.0
        # empty input
        e = sc.trace_rays(mid, np.zeros((0, 3), f32), np.zeros((0, 3), f32))
        assert len(e["prim"]) == 0
        sc.close()


def test_rejects_malformed_input(ctx, desc):
    import hpt
    m = desc.meshes[0]
    sc = hpt.Scene(ctx)
    nodes = host_tree(m)
    bad = nodes.copy(); bad["right"][0] = 0                      # cycle
    with pytest.raises(hpt.HptError, match="child index"):
        sc.add_mesh(m, bad)
    bad = nodes.copy(); bad["prim"][bad["prim"] >= 0] = 0        # duplicated primitive
    with pytest.raises(hpt.HptError, match="two leaves"):
        sc.add_mesh(m, bad)
    with pytest.raises(hpt.HptError, match="node count"):
        sc.add_mesh(m, nodes[:-2])
    import copy
    m2 = copy.deepcopy(m); m2.tris = m2.tris.copy(); m2.tris[5, 1] = 99999
    with pytest.raises(hpt.HptError, match="out of range"):
        sc.add_mesh(m2, None)
    with pytest.raises(hpt.HptError, match="not committed"):
        sc.trace_rays(0, np.zeros((1, 3), np.float32), np.zeros((1, 3), np.float32))
    with pytest.raises(hpt.HptError, match="singular"):
        sc.add_node(-1, 0, -1, -1, np.zeros(9, np.float32), np.zeros(3, np.float32))
    # scene.rs:330 divides by width and height: an image with exactly one zero dimension panics in the reference
    import ctypes as C
    px = np.zeros(15, np.uint8)
    for w, h in ((0, 5), (5, 0)):
        assert hpt.lib.hpt_scene_add_image(sc.h, px.ctypes.data_as(hpt.U8P), w, h, None) == -1 and b"zero" in hpt.lib.hpt_last_error()
    assert hpt.lib.hpt_scene_add_image(sc.h, None, 0, 0, None) == 0      # 0 x 0: the reference's "no data" image (black)
    sc.close()


def test_context_destroyed_before_its_scene(desc):
    """Python's __del__ order at interpreter exit (or a host that frees the context first): the context must stay alive
    until its last scene is gone -- hpt_context_destroy only marks it then."""
    import hpt
    c = hpt.Context(0)
    sc = hpt.Scene(c)
    mid = sc.add_mesh(desc.meshes[0], None)
    sc.commit()
    c.close()                                         # context first
    org = np.array([[0, 0, 50]], np.float32); d = np.array([[0, 0, -1]], np.float32)
    assert sc.trace_rays(mid, org, d)["prim"][0] >= 0 # the scene still works: streams and device are alive
    sc.close()                                        # frees the context with it
