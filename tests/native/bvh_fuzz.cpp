#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
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
    std::printf("accepted %d rejected %d\n", ok, rej);
    return 0;
}
