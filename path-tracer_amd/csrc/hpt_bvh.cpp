// hpt_bvh.cpp -- see hpt_bvh.h.  Host only; no recursion (reference trees can be O(n) deep).
#include "hpt_bvh.h"
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <thread>
#include <cstring>
#include <limits>

namespace {
struct Box {
    float mn[3], mx[3];
    void reset() { for (int k = 0; k < 3; ++k) { mn[k] = std::numeric_limits<float>::infinity(); mx[k] = -std::numeric_limits<float>::infinity(); } }
    void grow(const float* p) { for (int k = 0; k < 3; ++k) { mn[k] = std::min(mn[k], p[k]); mx[k] = std::max(mx[k], p[k]); } }
    void grow(const Box& b) { for (int k = 0; k < 3; ++k) { mn[k] = std::min(mn[k], b.mn[k]); mx[k] = std::max(mx[k], b.mx[k]); } }
    float half_area() const {
        float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        return dx * dy + dy * dz + dz * dx;
    }
};
Box tri_box(const float* verts, const uint32_t* t) {   // geometry.rs:231-233
    Box b; b.reset();
    b.grow(verts + 3 * (size_t)t[0]); b.grow(verts + 3 * (size_t)t[1]); b.grow(verts + 3 * (size_t)t[2]);
    return b;
}
}  // namespace

bool tree_from_reference_nodes(const hpt_bvh_node* nodes, int32_t nnodes, int32_t ntri, HostTree& out, std::string& err) {
    if (ntri <= 0) { err = "Cannot construct empty BVH"; return false; }   // bvh.rs:37
    if (nnodes != 2 * ntri - 1) { err = "bvh: node count must be 2*ntri-1 (one primitive per leaf)"; return false; }
    out.left.resize(nnodes); out.right.resize(nnodes); out.prim.resize(nnodes); out.root = 0;
    std::vector<uint8_t> seen_prim((size_t)ntri, 0), visited((size_t)nnodes, 0);
    for (int32_t i = 0; i < nnodes; ++i) {
        const hpt_bvh_node& n = nodes[i];
        if (n.prim >= 0) {
            if (n.prim >= ntri) { err = "bvh: leaf primitive index out of range"; return false; }
            if (seen_prim[n.prim]) { err = "bvh: primitive referenced by two leaves"; return false; }
            seen_prim[n.prim] = 1;
            out.left[i] = out.right[i] = -1; out.prim[i] = n.prim;
        } else {
            // pre-order layout (bvh.rs:111-136): children come after their parent => the structure is acyclic
            if (n.left <= i || n.left >= nnodes || n.right <= i || n.right >= nnodes || n.left == n.right) { err = "bvh: child index out of order/range"; return false; }
            out.left[i] = n.left; out.right[i] = n.right; out.prim[i] = -1;
        }
    }
    // every node must be reachable exactly once from the root
    std::vector<int32_t> stack{0};
    int64_t reached = 0;
    while (!stack.empty()) {
        int32_t i = stack.back(); stack.pop_back();
        if (visited[i]) { err = "bvh: node has two parents"; return false; }
        visited[i] = 1; ++reached;
        if (out.prim[i] < 0) { stack.push_back(out.left[i]); stack.push_back(out.right[i]); }
    }
    if (reached != nnodes) { err = "bvh: unreachable nodes"; return false; }
    return true;
}

void tree_build_sah(const float* verts, const uint32_t* tri_idx, int32_t ntri, HostTree& out) {
    std::vector<float> b((size_t)ntri * 6);
    for (int32_t i = 0; i < ntri; ++i) {
        const Box x = tri_box(verts, tri_idx + 3 * (size_t)i);
        for (int k = 0; k < 3; ++k) { b[6 * (size_t)i + k] = x.mn[k]; b[6 * (size_t)i + 3 + k] = x.mx[k]; }
    }
    tree_build_sah_boxes(b.data(), ntri, out);
}

// Binned SAH (16 bins; all three axes up to 4096 primitives, the longest axis above), one primitive per leaf.
// A subtree over n primitives owns exactly 2n-1 consecutive node slots (its root first, then the left subtree, then
// the right one), so subtrees are independent: the top of the tree is split on the calling thread until the ranges are
// small, the ranges below are built by a pool of threads.  The result does not depend on the number of threads.
namespace {
struct SahBuilder {
    static constexpr int NB = 16;
    const std::vector<Box>& boxes; const std::vector<float>& cent; std::vector<int32_t>& ids; HostTree& out;
    struct Job { int32_t node, lo, hi; };

    // splits [lo, hi) (n >= 2) and returns the split position
    int32_t split(int32_t lo, int32_t hi) const {
        const int32_t n = hi - lo;
        Box cb; cb.reset();
        for (int32_t i = lo; i < hi; ++i) cb.grow(&cent[3 * (size_t)ids[i]]);
        int axis = 0;
        const float ext[3] = {cb.mx[0] - cb.mn[0], cb.mx[1] - cb.mn[1], cb.mx[2] - cb.mn[2]};
        if (ext[1] > ext[axis]) axis = 1;
        if (ext[2] > ext[axis]) axis = 2;
        int32_t mid = lo + n / 2;
        if (!(ext[axis] > 0.0f)) return mid;
        int best_axis = -1, best_split = -1; float best_cost = std::numeric_limits<float>::infinity();
        for (int ax = 0; ax < 3; ++ax) {
            if (!(ext[ax] > 0.0f)) continue;
            if (n > 4096 && ax != axis) continue;   // large ranges: longest axis only
            Box bb[NB]; int32_t cnt[NB];
            for (int b = 0; b < NB; ++b) { bb[b].reset(); cnt[b] = 0; }
            const float scale = (float)NB / ext[ax];
            for (int32_t i = lo; i < hi; ++i) {
                int b = (int)((cent[3 * (size_t)ids[i] + ax] - cb.mn[ax]) * scale);
                b = std::min(std::max(b, 0), NB - 1);
                bb[b].grow(boxes[ids[i]]); cnt[b]++;
            }
            float right_area[NB]; int32_t right_cnt[NB];
            Box acc; acc.reset(); int32_t c = 0;
            for (int b = NB - 1; b > 0; --b) { acc.grow(bb[b]); c += cnt[b]; right_area[b] = c ? acc.half_area() : 0.0f; right_cnt[b] = c; }
            acc.reset(); c = 0;
            for (int b = 0; b < NB - 1; ++b) {
                acc.grow(bb[b]); c += cnt[b];
                if (c == 0 || right_cnt[b + 1] == 0) continue;
                const float cost = acc.half_area() * (float)c + right_area[b + 1] * (float)right_cnt[b + 1];
                if (cost < best_cost) { best_cost = cost; best_axis = ax; best_split = b; }
            }
        }
        if (best_axis >= 0) {
            const float scale = (float)NB / ext[best_axis];
            const float mnv = cb.mn[best_axis];
            auto it = std::partition(ids.begin() + lo, ids.begin() + hi, [&](int32_t id) {
                int b = (int)((cent[3 * (size_t)id + best_axis] - mnv) * scale);
                b = std::min(std::max(b, 0), NB - 1);
                return b <= best_split;
            });
            const int32_t m = (int32_t)(it - ids.begin());
            if (m > lo && m < hi) mid = m;
        }
        return mid;
    }
    // one step: leaf, or split + children in their fixed slots; pushes the child jobs
    void step(const Job& j, std::vector<Job>& todo) const {
        const int32_t n = j.hi - j.lo;
        if (n == 1) { out.prim[j.node] = ids[j.lo]; return; }
        const int32_t mid = split(j.lo, j.hi);
        const int32_t l = j.node + 1, r = j.node + 2 * (mid - j.lo);   // the left subtree owns 2*(mid-lo)-1 slots
        out.left[j.node] = l; out.right[j.node] = r;
        todo.push_back({r, mid, j.hi});
        todo.push_back({l, j.lo, mid});
    }
    void run(Job root) const { std::vector<Job> todo{root}; while (!todo.empty()) { const Job j = todo.back(); todo.pop_back(); step(j, todo); } }
};
int builder_threads() {
    int n = (int)std::thread::hardware_concurrency();
    if (const char* e = getenv("OMP_NUM_THREADS")) { const int v = atoi(e); if (v > 0) n = std::min(n > 0 ? n : v, v); }
    return std::max(1, std::min(n, 16));
}
}  // namespace

void tree_build_sah_boxes(const float* leaf_boxes, int32_t ntri, HostTree& out) {
    std::vector<Box> boxes((size_t)ntri);
    std::vector<float> cent((size_t)ntri * 3);
    for (int32_t i = 0; i < ntri; ++i) {
        for (int k = 0; k < 3; ++k) { boxes[i].mn[k] = leaf_boxes[6 * (size_t)i + k]; boxes[i].mx[k] = leaf_boxes[6 * (size_t)i + 3 + k]; }
        for (int k = 0; k < 3; ++k) cent[3 * (size_t)i + k] = 0.5f * (boxes[i].mn[k] + boxes[i].mx[k]);
    }
    std::vector<int32_t> ids((size_t)ntri);
    for (int32_t i = 0; i < ntri; ++i) ids[i] = i;
    const size_t nn = 2 * (size_t)ntri - 1;
    out.left.assign(nn, -1); out.right.assign(nn, -1); out.prim.assign(nn, -1);
    out.root = 0;
    const SahBuilder B{boxes, cent, ids, out};
    const int nthreads = ntri >= (1 << 16) ? builder_threads() : 1;
    if (nthreads == 1) { B.run({0, 0, ntri}); return; }
    // top of the tree on this thread, down to ranges of about ntri / (8 * threads) primitives
    const int32_t grain = std::max(4096, ntri / (8 * nthreads));
    std::vector<SahBuilder::Job> todo{{0, 0, ntri}}, tasks;
    while (!todo.empty()) {
        const SahBuilder::Job j = todo.back(); todo.pop_back();
        if (j.hi - j.lo <= grain) tasks.push_back(j); else B.step(j, todo);
    }
    std::atomic<size_t> next{0};
    std::vector<std::thread> pool;
    for (int t = 0; t < nthreads; ++t)
        pool.emplace_back([&]() { for (size_t k; (k = next.fetch_add(1)) < tasks.size();) B.run(tasks[k]); });
    for (std::thread& th : pool) th.join();
}

bool flatten_tree(const HostTree& tree, const float* leaf_boxes, int32_t nleaf, FlatMesh& out, std::string& err) {
    const int32_t nn = (int32_t)tree.left.size();
    (void)nleaf;
    std::vector<int32_t> order; order.reserve(nn);
    {
        std::vector<int32_t> st{tree.root};
        while (!st.empty()) {
            int32_t i = st.back(); st.pop_back();
            order.push_back(i);
            if (tree.prim[i] < 0) { st.push_back(tree.left[i]); st.push_back(tree.right[i]); }
        }
    }
    if ((int32_t)order.size() != nn) { err = "bvh: tree does not cover all nodes"; return false; }
    std::vector<Box> box((size_t)nn);
    for (int32_t k = nn - 1; k >= 0; --k) {
        const int32_t i = order[k];
        if (tree.prim[i] >= 0) {   // tight leaf box (bvh.rs:50-53)
            const float* b = leaf_boxes + 6 * (size_t)tree.prim[i];
            for (int c = 0; c < 3; ++c) { box[i].mn[c] = b[c]; box[i].mx[c] = b[3 + c]; }
        } else { box[i] = box[tree.left[i]]; box[i].grow(box[tree.right[i]]); }   // tight refit (drops quirk 2)
    }
    // breadth-first numbering of interior nodes + depth
    std::vector<int32_t> bfs_index((size_t)nn, -1), depth((size_t)nn, 0), queue;
    out.n_interior = 0; out.max_depth = 0;
    if (tree.prim[tree.root] < 0) { queue.push_back(tree.root); depth[tree.root] = 1; }
    for (size_t h = 0; h < queue.size(); ++h) {
        const int32_t i = queue[h];
        bfs_index[i] = out.n_interior++;
        out.max_depth = std::max(out.max_depth, depth[i]);
        for (int32_t c : {tree.left[i], tree.right[i]})
            if (tree.prim[c] < 0) { depth[c] = depth[i] + 1; queue.push_back(c); }
    }
    auto ref_of = [&](int32_t i) -> int32_t { return tree.prim[i] >= 0 ? ~tree.prim[i] : bfs_index[i]; };
    out.root_ref = ref_of(tree.root);
    out.nodes.assign((size_t)out.n_interior * 16, 0.0f);
    for (int32_t i : queue) {
        float* q = &out.nodes[(size_t)bfs_index[i] * 16];
        const Box& L = box[tree.left[i]]; const Box& R = box[tree.right[i]];
        const int32_t lr = ref_of(tree.left[i]), rr = ref_of(tree.right[i]);
        q[0] = L.mn[0]; q[1] = L.mn[1]; q[2] = L.mn[2]; std::memcpy(&q[3], &lr, 4);
        q[4] = L.mx[0]; q[5] = L.mx[1]; q[6] = L.mx[2]; std::memcpy(&q[7], &rr, 4);
        q[8] = R.mn[0]; q[9] = R.mn[1]; q[10] = R.mn[2];
        q[12] = R.mx[0]; q[13] = R.mx[1]; q[14] = R.mx[2];
    }
    for (int k = 0; k < 3; ++k) { out.bmin[k] = box[tree.root].mn[k]; out.bmax[k] = box[tree.root].mx[k]; }
    return true;
}

bool flatten_mesh(const HostTree& tree, const float* verts, const uint32_t* tri_idx, const float* normals, const uint32_t* nrm_idx,
                  const float* uvw, const uint32_t* uv_idx, int32_t ntri, FlatMesh& out, std::string& err) {
    std::vector<float> leaf((size_t)ntri * 6);
    for (int32_t f = 0; f < ntri; ++f) {
        const Box x = tri_box(verts, tri_idx + 3 * (size_t)f);
        for (int k = 0; k < 3; ++k) { leaf[6 * (size_t)f + k] = x.mn[k]; leaf[6 * (size_t)f + 3 + k] = x.mx[k]; }
    }
    if (!flatten_tree(tree, leaf.data(), ntri, out, err)) return false;
    out.tris.assign((size_t)ntri * 12, 0.0f);
    out.trishade.assign((size_t)ntri * 28, 0.0f);
    for (int32_t f = 0; f < ntri; ++f) {
        const float* a = verts + 3 * (size_t)tri_idx[3 * (size_t)f];
        const float* b = verts + 3 * (size_t)tri_idx[3 * (size_t)f + 1];
        const float* c = verts + 3 * (size_t)tri_idx[3 * (size_t)f + 2];
        float* t = &out.tris[(size_t)f * 12];
        for (int k = 0; k < 3; ++k) { t[k] = a[k]; t[4 + k] = b[k] - a[k]; t[8 + k] = c[k] - a[k]; }   // geometry.rs:189-190
        float* s = &out.trishade[(size_t)f * 28];
        for (int k = 0; k < 3; ++k) { s[k] = a[k]; s[3 + k] = b[k]; s[6 + k] = c[k]; }
        for (int v = 0; v < 3; ++v) {
            const float* n = normals + 3 * (size_t)nrm_idx[3 * (size_t)f + v];
            const float* u = uvw + 3 * (size_t)uv_idx[3 * (size_t)f + v];
            for (int k = 0; k < 3; ++k) { s[9 + 3 * v + k] = n[k]; s[18 + 3 * v + k] = u[k]; }
        }
    }
    return true;
}

// ------------------------------------------------------------------------------------------------ Node32
QuantFrame quant_frame(const FlatMesh& fm) {
    QuantFrame f;
    double ext[3], maxext = 0.0;
    for (int k = 0; k < 3; ++k) { ext[k] = (double)fm.bmax[k] - (double)fm.bmin[k]; maxext = std::max(maxext, ext[k]); }
    f.quantum = 0.0;
    for (int k = 0; k < 3; ++k) {
        // coordinates of the tree map to [8, 65527]: room for the outward rounding without clamping
        const double e = std::max({ext[k], 1e-5 * maxext, 1e-20});
        const double s = e / (65535.0 - 16.0);
        f.scale[k] = (float)s;
        if ((double)f.scale[k] < s) f.scale[k] = std::nextafterf(f.scale[k], INFINITY);   // never shrink the grid
        f.base[k] = (float)((double)fm.bmin[k] - 8.0 * (double)f.scale[k]);
        if ((double)f.base[k] > (double)fm.bmin[k] - 8.0 * (double)f.scale[k]) f.base[k] = std::nextafterf(f.base[k], -INFINITY);
        f.quantum = std::max(f.quantum, (double)f.scale[k]);
    }
    // mean extent (largest axis) of the leaf boxes = the child boxes that belong to triangle refs
    double sum = 0.0; size_t n = 0;
    for (int32_t i = 0; i < fm.n_interior; ++i) {
        const float* q = &fm.nodes[(size_t)i * 16];
        int32_t refs[2]; std::memcpy(&refs[0], &q[3], 4); std::memcpy(&refs[1], &q[7], 4);
        for (int c = 0; c < 2; ++c) if (refs[c] < 0) {
            const float* mn = q + 8 * c; const float* mx = q + 8 * c + 4;
            sum += std::max({(double)mx[0] - mn[0], (double)mx[1] - mn[1], (double)mx[2] - mn[2]}); ++n;
        }
    }
    f.mean_leaf_extent = n ? sum / (double)n : maxext;
    return f;
}

bool quantize_nodes(const float* nodes64, int32_t n_interior, const QuantFrame& f, std::vector<uint32_t>& out, std::string& err) {
    out.assign((size_t)n_interior * 8, 0u);
    for (int32_t i = 0; i < n_interior; ++i) {
        const float* q = nodes64 + (size_t)i * 16;
        uint32_t* w = &out[(size_t)i * 8];
        std::memcpy(&w[3], &q[3], 4); std::memcpy(&w[7], &q[7], 4);
        for (int c = 0; c < 2; ++c) {
            const float* mn = q + 8 * c; const float* mx = q + 8 * c + 4;
            for (int k = 0; k < 3; ++k) {
                if (!(mn[k] <= mx[k])) { err = "bvh: empty or NaN child box cannot be quantised"; return false; }
                const double lo = std::floor(((double)mn[k] - (double)f.base[k]) / (double)f.scale[k]) - 1.0;
                const double hi = std::ceil(((double)mx[k] - (double)f.base[k]) / (double)f.scale[k]) + 1.0;
                if (lo < 0.0 || hi > 65535.0) { err = "bvh: child box outside the quantisation grid"; return false; }
                const uint32_t ql = (uint32_t)lo, qh = (uint32_t)hi;
                // what the device computes at worst: an f32 fma of the same decode
                if (std::fmaf((float)ql, f.scale[k], f.base[k]) > mn[k] || std::fmaf((float)qh, f.scale[k], f.base[k]) < mx[k]) {
                    err = "bvh: quantised child box does not contain the original"; return false;
                }
                w[4 * c + k] = ql | (qh << 16);
            }
        }
    }
    return true;
}
