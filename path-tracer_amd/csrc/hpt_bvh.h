// hpt_bvh.h -- host side of the BVH hand-over: validate the reference-built tree (bvh.rs:35-146 output), or build
// a binned-SAH tree when the host gives none, then flatten either to the Node64 / Tri48 / TriShade device layout.
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include "../../include/hpt.h"
#ifndef HPT_TRI_QUADS
#define HPT_TRI_QUADS 4          // 16-B quads per triangle record: see hpt_device.h
#endif
#ifndef HPT_TRI_FLOATS
#define HPT_TRI_FLOATS (HPT_TRI_QUADS * 4)
#endif

struct HostTree {           // binary tree over primitives, one primitive per leaf (bvh.rs:49-55)
    std::vector<int32_t> left, right, prim;   // interior: left/right >= 0, prim = -1; leaf: prim >= 0
    int32_t root = 0;
};

struct FlatMesh {
    std::vector<float> nodes;      // n_interior * 16 floats (Node64)
    std::vector<float> tris;       // ntri * HPT_TRI_FLOATS floats (Tri48)
    std::vector<float> trishade;   // ntri * 28 floats (TriShade)
    int32_t root_ref = -1;         // LOCAL ref: >= 0 interior index, < 0 ~local triangle
    int32_t n_interior = 0;
    int32_t max_depth = 0;         // deepest interior level (root = 1); 0 for a one-triangle mesh
    float bmin[3], bmax[3];
};

// Copies and validates the reference-layout node array. Returns false and sets err on malformed input.
bool tree_from_reference_nodes(const hpt_bvh_node* nodes, int32_t nnodes, int32_t ntri, HostTree& out, std::string& err);
// Binned SAH over triangle boxes/centroids (used when the host hands no tree, and for synthetic large scenes).
void tree_build_sah(const float* verts, const uint32_t* tri_idx, int32_t ntri, HostTree& out);
// same over arbitrary leaf boxes (n x 6: min, max) -- the TLAS over instance world boxes uses this
void tree_build_sah_boxes(const float* leaf_boxes, int32_t n, HostTree& out);
// Restructures a tree over the same leaves for fewer box steps per ray: (rebuild) top-down SAH re-partition, then
// `passes` rounds of insertion-based optimisation.  A result deeper than hard_depth (the traversal stack limit) is
// never produced; the insertion rounds are dropped when they leave the tree deeper than max(soft_depth, depth before).
// Returns whether the tree changed.  Hits do not depend on the tree, so this never changes a result.
bool tree_optimize(HostTree& tree, const float* leaf_boxes, int32_t nleaf, int32_t hard_depth, int32_t soft_depth, int passes, bool rebuild);
// sum of the interior nodes' surface areas / the root's: expected box steps of a random ray through the root box
double tree_sah_cost(const HostTree& tree, const float* leaf_boxes);
int32_t tree_interior_depth(const HostTree& tree);
// tight refit + breadth-first renumbering + Node64 packing of any tree over leaf boxes; leaf refs are ~prim
bool flatten_tree(const HostTree& tree, const float* leaf_boxes, int32_t nleaf, FlatMesh& out, std::string& err);
// Tight refit + breadth-first renumbering + packing.  Child refs are LOCAL to the mesh; the caller rebases them.
bool flatten_mesh(const HostTree& tree, const float* verts, const uint32_t* tri_idx, const float* normals, const uint32_t* nrm_idx,
                  const float* uvw, const uint32_t* uv_idx, int32_t ntri, FlatMesh& out, std::string& err);

// ---- Node32: the same tree with 16-bit child boxes on a per-tree grid (2 quads per node instead of 4) ----
// word layout of one node (8 x uint32): {Lx, Ly, Lz, refL, Rx, Ry, Rz, refR} with each box word = min | max << 16.
// A coordinate decodes to qbase + q * qscale; boxes only ever GROW (min rounded down, max rounded up, one more
// quantum on each side for the f32 decode), so culling stays conservative and hits are unchanged.
struct QuantFrame { float base[3], scale[3]; double mean_leaf_extent, quantum; };
// grid over [bmin, bmax] of the tree; mean_leaf_extent / quantum tell the caller whether 16 bits resolve the leaves
QuantFrame quant_frame(const FlatMesh& fm);
// nodes64: n_interior * 16 floats (refs already in place) -> n_interior * 8 words; false if a box fails the decode check
bool quantize_nodes(const float* nodes64, int32_t n_interior, const QuantFrame& f, std::vector<uint32_t>& out, std::string& err);
