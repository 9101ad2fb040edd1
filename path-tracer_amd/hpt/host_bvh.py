"""Host-side BVH build, mirroring the part of the reference that stays on the host (north_star: "bvh.rs still builds
the tree on the host"): Mesh::build_bvh (geometry.rs:223-239) + BVH::build / split (bvh.rs:35-146), both quirks kept
(half-extent "center" bvh.rs:68, parent box handed to the right child bvh.rs:144).  The result is the pre-order
node array that crosses the C ABI as hpt_bvh_node[] (include/hpt.h); the library refits and flattens it.

Pure Python/numpy in f32; meant for reference-sized meshes (the teapot: 6 320 triangles, ~0.3 s).  For large synthetic
meshes pass nodes=None to hpt_scene_add_mesh and the library builds a SAH tree instead.
"""
import sys

import numpy as np

NODE_DTYPE = np.dtype([("left", "i4"), ("right", "i4"), ("prim", "i4"), ("p1", "f4", 3), ("p2", "f4", 3)])


def mesh_box(verts):
    """load.rs:284-293"""
    v = np.asarray(verts, dtype=np.float32)
    return v.min(axis=0), v.max(axis=0)


def triangle_boxes(verts, tris):
    """geometry.rs:226-236"""
    v = np.asarray(verts, dtype=np.float32)[np.asarray(tris, dtype=np.int64)]   # ntri x 3 x 3
    return v.min(axis=1), v.max(axis=1)


def build_reference_bvh(verts, tris):
    """BVH::build (bvh.rs:35) over the per-triangle boxes with the mesh box as root box. Returns NODE_DTYPE[2n-1]."""
    p1, p2 = triangle_boxes(verts, tris)
    n = len(p1)
    if n == 0:
        raise ValueError("Cannot construct empty BVH")   # bvh.rs:37
    root1, root2 = mesh_box(verts)
    nodes = np.zeros(2 * n - 1, dtype=NODE_DTYPE)
    count = 0
    half = ((p2 - p1) / np.float32(2.0)).astype(np.float32)   # the reference's "center" (bvh.rs:68,82,96)
    elems = np.arange(n, dtype=np.int64)
    # explicit stack of (lo, hi, box1, box2, parent, is_right); left subtree is emitted before the right one
    stack = [(0, n, root1, root2, -1, False)]
    while stack:
        lo, hi, b1, b2, parent, is_right = stack.pop()
        idx = count
        count += 1
        if parent >= 0 and is_right:
            nodes["right"][parent] = idx
        if hi - lo == 1:
            e = elems[lo]
            nodes[idx] = (-1, -1, e, p1[e], p2[e])
            continue
        size = (b2 - b1).astype(np.float32)
        if size[0] > size[1] and size[0] > size[2]:
            ax = 0
        elif size[1] > size[2]:
            ax = 1
        else:
            ax = 2
        pivot = np.float32(b1[ax] + np.float32(size[ax] / np.float32(2.0)))
        seg = elems[lo:hi]
        # the Lomuto-style swap loop of bvh.rs:66-76, vectorised: elements with center < pivot move to the front in
        # order; the displaced ones follow in the order the swaps leave them
        mask = half[seg, ax] < pivot
        j = int(mask.sum())
        if 0 < j < hi - lo:
            seg = _lomuto(seg, mask)
            elems[lo:hi] = seg
        if j == 0 or j == hi - lo:
            j = (hi - lo) // 2
        nodes[idx] = (idx + 1, 0, -1, b1, b2)
        left = elems[lo:lo + j]
        l1 = p1[left].min(axis=0)
        l2 = p2[left].max(axis=0)
        stack.append((lo + j, hi, b1, b2, idx, True))     # right child gets the PARENT's box (bvh.rs:144)
        stack.append((lo, lo + j, l1, l2, idx, False))    # left child gets the tight union (bvh.rs:119-125)
    assert count == 2 * n - 1
    return nodes


def _lomuto(seg, mask):
    """Exact element order produced by the swap loop `if center < pivot { swap(elems[i], elems[j]); j += 1 }`."""
    seg = seg.copy()
    j = 0
    for i in np.nonzero(mask)[0]:
        if i != j:
            seg[i], seg[j] = seg[j], seg[i]
        j += 1
    return seg
