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
    static constexpr int32_t SWEEP_MAX = 4096;
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
        if (n <= SWEEP_MAX) {      // exact sweep: every split position of every axis
            std::vector<int32_t> tmp(ids.begin() + lo, ids.begin() + hi), best_order;
            std::vector<float> right_area((size_t)n);
            float best_cost = std::numeric_limits<float>::infinity(); int32_t best_i = -1;
            for (int ax = 0; ax < 3; ++ax) {
                if (!(ext[ax] > 0.0f)) continue;
                std::sort(tmp.begin(), tmp.end(), [&](int32_t a, int32_t b) {
                    const float ca = cent[3 * (size_t)a + ax], cb2 = cent[3 * (size_t)b + ax];
                    return ca != cb2 ? ca < cb2 : a < b;
                });
                Box acc; acc.reset();
                for (int32_t i = n - 1; i > 0; --i) { acc.grow(boxes[tmp[i]]); right_area[i] = acc.half_area(); }
                acc.reset();
                bool improved = false;
                for (int32_t i = 1; i < n; ++i) {
                    acc.grow(boxes[tmp[i - 1]]);
                    const float cost = acc.half_area() * (float)i + right_area[i] * (float)(n - i);
                    if (cost < best_cost) { best_cost = cost; best_i = i; improved = true; }
                }
                if (improved) best_order = tmp;
            }
            if (best_i > 0) { std::copy(best_order.begin(), best_order.end(), ids.begin() + lo); mid = lo + best_i; }
            return mid;
        }
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

// ------------------------------------------------------------------------------------------------ tree optimiser
// The tree the host hands over is the one bvh.rs:48-146 builds: on the shipped teapot only 69 of its 6 319 splits are
// real partitions (the others fall back to the index median), so it costs 6.5 box steps per ray where a
// surface-area-heuristic tree over the same leaves costs 3.  Hits do not depend on the tree (closest t, ties to the
// lower triangle index), so hpt_scene_commit restructures what it flattens: (1) a top-down SAH re-partition of the
// host's leaves -- exact sweep over all three axes for ranges up to 4 096 leaves, binned above -- and (2) rounds of
// insertion-based optimisation (remove a subtree, re-insert it where the surface area of the tree grows least,
// branch-and-bound search; Bittner, Hapala, Havran 2013).  bvh.rs stays the builder on the host; this repairs its
// output the way the tight refit already repaired its boxes.
namespace {
inline Box leaf_box(const float* leaf_boxes, int32_t p) {
    Box b; const float* q = leaf_boxes + 6 * (size_t)p;
    for (int k = 0; k < 3; ++k) { b.mn[k] = q[k]; b.mx[k] = q[3 + k]; }
    return b;
}
inline Box box_union(const Box& a, const Box& b) { Box r = a; r.grow(b); return r; }

struct Reinserter {
    HostTree& t; std::vector<int32_t> parent, ih; std::vector<Box> box;      // ih: interior levels in the subtree (leaf 0)
    int32_t depth_limit = 1 << 30;
    Reinserter(HostTree& tree, const float* leaf_boxes) : t(tree) {
        const int32_t nn = (int32_t)t.left.size();
        parent.assign((size_t)nn, -1); ih.assign((size_t)nn, 0); box.resize((size_t)nn);
        std::vector<int32_t> order; order.reserve((size_t)nn);
        std::vector<int32_t> st{t.root};
        while (!st.empty()) {
            const int32_t i = st.back(); st.pop_back(); order.push_back(i);
            if (t.prim[i] < 0) { parent[t.left[i]] = i; parent[t.right[i]] = i; st.push_back(t.left[i]); st.push_back(t.right[i]); }
        }
        for (size_t k = order.size(); k-- > 0;) {
            const int32_t i = order[k];
            box[i] = t.prim[i] >= 0 ? leaf_box(leaf_boxes, t.prim[i]) : box_union(box[t.left[i]], box[t.right[i]]);
            ih[i] = t.prim[i] >= 0 ? 0 : 1 + std::max(ih[t.left[i]], ih[t.right[i]]);
        }
    }
    void refit_up(int32_t i) {
        for (; i >= 0; i = parent[i]) {
            const Box nb = box_union(box[t.left[i]], box[t.right[i]]);
            const int32_t nh = 1 + std::max(ih[t.left[i]], ih[t.right[i]]);
            if (nh == ih[i] && std::memcmp(&nb, &box[i], sizeof nb) == 0) break;
            box[i] = nb; ih[i] = nh;
        }
    }
    double cost() const {
        double c = 0.0;
        for (size_t i = 0; i < box.size(); ++i) if (t.prim[i] < 0) c += box[i].half_area();
        return c;
    }
    struct Cand { float induced; int32_t node, level; };
    // moves subtree n to the position where the tree's surface area grows least; returns true if the tree changed
    bool reinsert(int32_t n, std::vector<Cand>& heap) {
        const int32_t p = parent[n];
        if (p < 0 || parent[p] < 0) return false;      // the root and its children stay (the root record is the entry point)
        const int32_t g = parent[p], s = t.left[p] == n ? t.right[p] : t.left[p];
        const float before = box[p].half_area();
        // unlink p: s takes its place under g
        (t.left[g] == p ? t.left[g] : t.right[g]) = s; parent[s] = g;
        refit_up(g);
        const Box nb = box[n]; const float na = nb.half_area();
        // branch and bound over insertion positions x: cost = area(x U n) + sum over x's ancestors of their growth
        float best = std::numeric_limits<float>::infinity(); int32_t best_x = -1;
        heap.clear(); heap.push_back({0.0f, t.root, 1});
        auto cmp = [](const Cand& a, const Cand& b) { return a.induced > b.induced; };
        while (!heap.empty()) {
            std::pop_heap(heap.begin(), heap.end(), cmp);
            const Cand c = heap.back(); heap.pop_back();
            if (c.induced + na >= best) break;
            const float direct = box_union(box[c.node], nb).half_area();
            const float total = c.induced + direct;
            // the new parent takes x's level; x and n go one level down: deepest interior level below = level + max(ih)
            if (total < best && c.level + std::max(ih[c.node], ih[n]) <= depth_limit) { best = total; best_x = c.node; }
            const float child_induced = total - box[c.node].half_area();
            if (t.prim[c.node] < 0 && child_induced + na < best) {
                heap.push_back({child_induced, t.left[c.node], c.level + 1}); std::push_heap(heap.begin(), heap.end(), cmp);
                heap.push_back({child_induced, t.right[c.node], c.level + 1}); std::push_heap(heap.begin(), heap.end(), cmp);
            }
        }
        int32_t x = best_x;
        // putting it back where it was costs `before` more than the unlinked tree: only move for a strict gain
        if (x < 0 || !(best < before) || x == s) x = s;
        const int32_t xp = parent[x];
        t.left[p] = x; t.right[p] = n; parent[x] = p; parent[n] = p; parent[p] = xp;
        if (xp < 0) t.root = p; else (t.left[xp] == x ? t.left[xp] : t.right[xp]) = p;
        box[p] = box_union(box[x], nb); ih[p] = 1 + std::max(ih[x], ih[n]);
        if (xp >= 0) refit_up(xp);
        return x != s;
    }
    int32_t depth() const {      // deepest interior level, root = 1 (what flatten_tree reports as max_depth)
        int32_t best = 0;
        std::vector<std::pair<int32_t, int32_t>> st{{t.root, 1}};
        while (!st.empty()) {
            const auto [i, d] = st.back(); st.pop_back();
            if (t.prim[i] >= 0) continue;
            best = std::max(best, d);
            st.push_back({t.left[i], d + 1}); st.push_back({t.right[i], d + 1});
        }
        return best;
    }
};
}  // namespace

double tree_sah_cost(const HostTree& tree, const float* leaf_boxes) {
    HostTree copy = tree;
    Reinserter R(copy, leaf_boxes);
    const double root = R.box[copy.root].half_area();
    return root > 0.0 ? R.cost() / root : 0.0;
}

int32_t tree_interior_depth(const HostTree& tree) {
    int32_t best = 0;
    std::vector<std::pair<int32_t, int32_t>> st{{tree.root, 1}};
    while (!st.empty()) {
        const auto [i, d] = st.back(); st.pop_back();
        if (tree.prim[i] >= 0) continue;
        best = std::max(best, d);
        st.push_back({tree.left[i], d + 1}); st.push_back({tree.right[i], d + 1});
    }
    return best;
}

bool tree_optimize(HostTree& tree, const float* leaf_boxes, int32_t nleaf, int32_t hard_depth, int32_t soft_depth, int passes, bool rebuild) {
    if (nleaf < 4) return false;
    bool changed = false;
    if (rebuild) {
        HostTree rebuilt;
        tree_build_sah_boxes(leaf_boxes, nleaf, rebuilt);
        if (tree_interior_depth(rebuilt) > hard_depth) return false;      // (median fallbacks keep it near log2 n)
        tree = rebuilt; changed = true;
    }
    if (passes <= 0) return changed;
    const int32_t depth_in = tree_interior_depth(tree);
    HostTree work = tree;
    Reinserter R(work, leaf_boxes);
    R.depth_limit = std::min(hard_depth, std::max(soft_depth, depth_in));
    const int32_t nn = (int32_t)work.left.size();
    std::vector<int32_t> order((size_t)nn);
    std::vector<Reinserter::Cand> heap;
    double prev = R.cost();
    const double first = prev;
    for (int pass = 0; pass < passes; ++pass) {
        for (int32_t i = 0; i < nn; ++i) order[i] = i;
        // largest boxes first: they are the ones whose misplacement costs most
        std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
            const float aa = R.box[a].half_area(), ab = R.box[b].half_area();
            return aa != ab ? aa > ab : a < b;
        });
        for (int32_t n : order) R.reinsert(n, heap);
        const double now = R.cost();
        if (!(now < prev * 0.999)) { prev = std::min(prev, now); break; }
        prev = now;
    }
    // a deeper tree needs more traversal-stack entries (LDS per lane): beyond soft_depth they cost resident workgroups,
    // which no few per cent of surface area win back -- keep the input then
    const int32_t depth_out = R.depth();
    if (prev < first && depth_out <= hard_depth && depth_out <= std::max(soft_depth, depth_in)) { tree = work; changed = true; }
    return changed;
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
    // Numbering of the interior nodes = their order in memory.  Breadth-first puts a node's record between those of its
    // cousins: a ray walking DOWN finds nothing it needs in the 128-B line a fetch brought in (4 Node32 / 2 Node64 records).
    // Treelet order: starting at a subtree root, up to 4 interior nodes -- the root, then repeatedly the frontier node with
    // the largest surface area, i.e. the one a ray that is here most probably visits next -- get consecutive indices, and
    // the subtrees hanging off the treelet follow depth-first, largest first, so a subtree stays contiguous at every scale
    // (L1 / L2 lines, Infinity Cache pages).  Traversal and hits do not depend on the numbering.  Measured (r02g): the
    // 1 M-triangle soup, whose 32 MB of nodes miss L2, k_extend -5.6 % (frame +3.6 %); trees that live in L2 +-0 (teapot,
    // 3 375 instanced teapots, the TLAS of 10 k spheres) -- so trees below 65 536 interior nodes keep the breadth-first
    // numbering.  HPT_NODE_ORDER=bfs / treelet forces one (A/B runs; the HPT_LDS_TOPCACHE variant of the kernels assumes bfs).
    {
        const char* e = getenv("HPT_NODE_ORDER");
        const bool treelet = e ? strcmp(e, "treelet") == 0 : out.n_interior >= 65536;
        if (treelet && out.n_interior > 4) {
            auto area = [&](int32_t i) { const Box& b = box[i]; const double dx = (double)b.mx[0] - b.mn[0], dy = (double)b.mx[1] - b.mn[1], dz = (double)b.mx[2] - b.mn[2]; return dx * dy + dy * dz + dz * dx; };
            std::vector<int32_t> roots{tree.root}, frontier;
            int32_t next = 0;
            while (!roots.empty()) {
                const int32_t r = roots.back(); roots.pop_back();
                frontier.clear();
                int32_t taken = 0, cur = r;
                for (;;) {
                    bfs_index[cur] = next++; ++taken;
                    for (int32_t c : {tree.left[cur], tree.right[cur]}) if (tree.prim[c] < 0) frontier.push_back(c);
                    if (taken == 4 || frontier.empty()) break;
                    size_t best = 0;
                    for (size_t k = 1; k < frontier.size(); ++k) if (area(frontier[k]) > area(frontier[best])) best = k;
                    cur = frontier[best]; frontier.erase(frontier.begin() + (ptrdiff_t)best);
                }
                // the rest of the frontier: roots of the next treelets, the largest on top of the stack (numbered next)
                std::sort(frontier.begin(), frontier.end(), [&](int32_t a, int32_t b) { return area(a) < area(b); });
                for (int32_t c : frontier) roots.push_back(c);
            }
        }
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
    out.tris.assign((size_t)ntri * HPT_TRI_FLOATS, 0.0f);
    out.trishade.assign((size_t)ntri * 28, 0.0f);
    for (int32_t f = 0; f < ntri; ++f) {
        const float* a = verts + 3 * (size_t)tri_idx[3 * (size_t)f];
        const float* b = verts + 3 * (size_t)tri_idx[3 * (size_t)f + 1];
        const float* c = verts + 3 * (size_t)tri_idx[3 * (size_t)f + 2];
        float* t = &out.tris[(size_t)f * HPT_TRI_FLOATS];
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
