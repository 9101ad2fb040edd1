#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <algorithm>
#include "hpt_bvh.h"
// builds a valid pre-order tree over n triangles, mutates it, and pushes it through validation + flatten
static int build(std::vector<hpt_bvh_node>& out, int lo, int hi, std::mt19937& rng) {
    int idx = (int)out.size(); out.push_back(hpt_bvh_node{});
    if (hi - lo == 1) { out[idx].left = out[idx].right = -1; out[idx].prim = lo; return idx; }
    int mid = lo + 1 + (int)(rng() % (unsigned)(hi - lo - 1));
    out[idx].prim = -1;
    int l = build(out, lo, mid, rng); int r = build(out, mid, hi, rng);
    out[idx].left = l; out[idx].right = r;
    return idx;
}
int main() {
    std::mt19937 rng(7);
    int ok = 0, rej = 0;
    for (int it = 0; it < 20000; ++it) {
        int n = 1 + (int)(rng() % 40);
        std::vector<float> verts(n * 9); for (auto& v : verts) v = (float)(rng() % 1000) / 100.0f;
        std::vector<uint32_t> tri(n * 3); for (int i = 0; i < n * 3; ++i) tri[i] = i;
        std::vector<hpt_bvh_node> nodes; build(nodes, 0, n, rng);
        int muts = (int)(rng() % 4);
        for (int m = 0; m < muts; ++m) {
            hpt_bvh_node& nd = nodes[rng() % nodes.size()];
            int* f = (rng() % 3 == 0) ? &nd.left : (rng() % 2 ? &nd.right : &nd.prim);
            switch (rng() % 6) { case 0: *f = -1; break; case 1: *f = 0; break; case 2: *f = (int)(rng() % (nodes.size() + 3)); break;
                                 case 3: *f = 0x7fffffff; break; case 4: *f = (int)0x80000000; break; default: *f = -(int)(rng() % 5); }
        }
        int nn = (int)nodes.size(); if (rng() % 10 == 0) nn = (int)(rng() % (nodes.size() + 2));
        HostTree t; std::string err;
        if (!tree_from_reference_nodes(nodes.data(), nn, n, t, err)) { ++rej; continue; }
        FlatMesh fm;
        std::vector<float> nrm(3, 0.f), uvw(3, 0.f); std::vector<uint32_t> z(n * 3, 0);
        if (!flatten_mesh(t, verts.data(), tri.data(), nrm.data(), z.data(), uvw.data(), z.data(), n, fm, err)) { ++rej; continue; }
        ++ok;
    }
    // SAH builder on degenerate inputs
    for (int n : {1, 2, 3, 7, 64, 1000}) {
        std::vector<float> verts(n * 9, 1.0f); std::vector<uint32_t> tri(n * 3); for (int i = 0; i < n * 3; ++i) tri[i] = i;
        HostTree t; tree_build_sah(verts.data(), tri.data(), n, t);
        FlatMesh fm; std::string err; std::vector<float> a(3, 0.f); std::vector<uint32_t> z(n * 3, 0);
        if (!flatten_mesh(t, verts.data(), tri.data(), a.data(), z.data(), a.data(), z.data(), n, fm, err)) { std::printf("sah flatten failed: %s\n", err.c_str()); return 1; }
        std::vector<float> nanv(n * 9, NAN);
        HostTree t2; tree_build_sah(nanv.data(), tri.data(), n, t2);
    }
    // tree optimiser: random trees over random boxes -> still a tree over exactly the same leaves, never costlier,
    // never deeper than the limits it was given
    for (int it = 0; it < 300; ++it) {
        int n = 4 + (int)(rng() % 400);
        std::vector<float> lb(n * 6);
        for (int i = 0; i < n; ++i) for (int k = 0; k < 3; ++k) {
            float c0 = (float)(rng() % 1000) / 10.0f, e = (it % 7 == 0) ? 0.0f : (float)(rng() % 100) / 20.0f;
            lb[6 * i + k] = c0; lb[6 * i + 3 + k] = c0 + e;
        }
        std::vector<hpt_bvh_node> nodes; build(nodes, 0, n, rng);
        HostTree t; std::string err;
        if (!tree_from_reference_nodes(nodes.data(), (int)nodes.size(), n, t, err)) { std::printf("valid tree rejected: %s\n", err.c_str()); return 1; }
        const double before = tree_sah_cost(t, lb.data());
        const int soft = 4 + (int)(rng() % 30), din = tree_interior_depth(t);
        HostTree o = t;
        const bool rebuild = rng() % 2;
        tree_optimize(o, lb.data(), n, 38, soft, (int)(rng() % 4), rebuild);
        std::vector<int> seen(n, 0); int leaves = 0, interior = 0;
        std::vector<int> st{o.root};
        while (!st.empty()) {
            int i = st.back(); st.pop_back();
            if (o.prim[i] >= 0) { seen[o.prim[i]]++; ++leaves; } else { ++interior; st.push_back(o.left[i]); st.push_back(o.right[i]); }
        }
        for (int i = 0; i < n; ++i) if (seen[i] != 1) { std::printf("optimiser lost or duplicated leaf %d\n", i); return 1; }
        if (leaves != n || interior != n - 1) { std::printf("optimiser: not a binary tree over the leaves\n"); return 1; }
        const double after = tree_sah_cost(o, lb.data());
        if (after > before * 1.0001 && !rebuild) { std::printf("insertion rounds raised the cost %g -> %g\n", before, after); return 1; }
        const int dout = tree_interior_depth(o);
        if (dout > 38 || (!rebuild && dout > std::max(soft, din))) { std::printf("optimiser depth %d (in %d, soft %d)\n", dout, din, soft); return 1; }
        FlatMesh fm;
        if (!flatten_tree(o, lb.data(), n, fm, err)) { std::printf("flatten after optimise: %s\n", err.c_str()); return 1; }
    }
    std::printf("accepted %d rejected %d\n", ok, rej);
    return 0;
}
