"""Classification of rays on which the GPU's closest hit names another node than the oracle's.

The GPU decides every hit with the reference's exact primitive arithmetic and culls with CONSERVATIVE boxes; the reference (and the
oracle, which follows it) culls with its own slab test (bvh.rs:225-281, geometry.rs:31), which rounds and is not conservative:
it can reject a box whose triangle the exact test hits.  So the only legitimate disagreement is
    "reference AABB false miss": the exact primitive test of the GPU's node, run WITHOUT any box test (tests/restatement2.py,
    brute force over every triangle of the mesh), hits at exactly the GPU's t, and the oracle reports nothing nearer.
Everything else -- the GPU missing or naming a farther hit than the oracle, a GPU hit the exact test does not reproduce, an equal-t
tie between nodes resolved differently -- fails the test that calls this."""
import numpy as np

import restatement2 as r2

F = np.float32


def _chain(desc, node_id):
    c = []
    while node_id >= 0:
        c.append(node_id); node_id = desc.nodes[node_id].parent
    return c[::-1]


def _exact_t(desc, node_id, o, d):
    """closest exact hit of ONE node's object for one world-space ray, no AABB test anywhere (scene.rs:228-232 per level, then
    geometry.rs:36-92 / 132-221 with the pre-test of geometry.rs:31 and the BVH left out)"""
    pos = np.asarray(o, F).reshape(1, 3); dr = np.asarray(d, F).reshape(1, 3)
    for nid in _chain(desc, node_id):
        nd = desc.nodes[nid]
        lp = r2.to_local_space(nd.m, nd.t, pos)
        dr = (r2.to_local_space(nd.m, nd.t, (pos + dr).astype(F)) - lp).astype(F)
        pos = lp
    keep = r2.box_intersect
    r2.box_intersect = lambda p1, p2, p, q: np.ones(len(p), bool)
    try:
        h = r2.geometry_intersect(desc, node_id, pos, dr)
    finally:
        r2.box_intersect = keep
    return h.t[0]


def classify_node_mismatches(desc, o, d, g, r, max_frac=5e-4):
    """asserts that every ray with g["node"] != r["node"] is a reference AABB false miss; returns their number"""
    bad = np.flatnonzero(g["node"] != r["node"])
    assert len(bad) <= max(3, int(max_frac * len(o))), "%d of %d rays disagree on the node" % (len(bad), len(o))
    for i in bad:
        assert g["node"][i] >= 0, "ray %d: the GPU misses, the oracle hits node %d at t = %r" % (i, r["node"][i], r["t"][i])
        if r["node"][i] >= 0:
            assert r["t"][i] != g["t"][i], "ray %d: equal-t tie between nodes %d / %d resolved differently" % (i, g["node"][i], r["node"][i])
            assert g["t"][i] < r["t"][i], "ray %d: the oracle's hit (node %d, t = %r) is nearer than the GPU's (node %d, t = %r)" % (
                i, r["node"][i], r["t"][i], g["node"][i], g["t"][i])
        t = _exact_t(desc, int(g["node"][i]), o[i], d[i])
        assert np.float32(t) == np.float32(g["t"][i]), "ray %d: GPU hit (node %d, t = %r) is not what the exact test gives (%r)" % (i, g["node"][i], g["t"][i], t)
    return len(bad)
