// hpt_group.cpp -- one call that uses every GPU of the node (include/hpt.h, hpt_group_*).
//
// The reference's one pixel loop saturates the whole machine (main.rs:57,72: nested rayon maps over pixels and samples).
// Here the machine is up to 8 MI355X: a group holds one context per GPU, the committed scene is REPLICATED on each
// (SURVEY 8e), every GPU renders a contiguous range of the samples of every pixel on a host thread of its own -- the
// counter-based RNG is keyed by (pixel, sample), so the set of paths does not depend on the number of GPUs -- and the one
// exchange step is a sum-reduce of the accumulators onto GPU 0: RCCL (`ncclReduce` of int64, communicator from
// `ncclCommInitAll`, over xGMI), or -- transport "p2p", chosen automatically when a device appears twice in the list,
// which RCCL refuses -- peer copies onto GPU 0 followed by adds.  The accumulators are 64-bit fixed point (include/hpt.h
// hpt_accum): integer sums do not depend on the order RCCL adds the ranks in, nor on how the samples were split, so the image
// is the one-GPU image bit for bit.
// Adaptive mode (main.rs:67-107) runs the rounds in lock step: each GPU renders its share of the round's samples, the
// per-channel min/max planes are combined with one all-reduce (max over [-min, max]), and every GPU then runs the
// contrast test on identical data and keeps an identical active set.
//
// RCCL is loaded with dlopen at group creation (librccl.so.1: the copy already in the process if there is one -- PyTorch
// bundles its own -- else /opt/rocm's), so a host that never creates a group does not depend on it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include "../../include/hpt.h"
#include "hpt_internal.h"
#include "hpt_launch.h"

#define GHIP(expr)                                                                                                       \
    do {                                                                                                                 \
        hipError_t e_ = (expr);                                                                                          \
        if (e_ != hipSuccess) return hpti_fail(HPT_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define GCHK(expr) do { int rc_ = (expr); if (rc_ != HPT_OK) return rc_; } while (0)

namespace {
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Rccl {
    void* lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclReduce) Reduce = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string where;
    int load() {
        if (lib) return HPT_OK;
        const char* names[] = {"librccl.so.1", "librccl.so"};
        for (const char* n : names) if (!lib) { lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (lib) where = std::string(n) + " (already loaded)"; }
        const char* paths[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
        for (const char* n : paths) if (!lib) { lib = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (lib) where = n; }
        if (!lib) return hpti_fail(HPT_ERR_DEVICE, "hpt_group_create: cannot load RCCL (librccl.so.1): %s", dlerror());
#define SYM(f) f = (decltype(f))dlsym(lib, "nccl" #f); if (!f) return hpti_fail(HPT_ERR_DEVICE, "hpt_group_create: RCCL has no nccl" #f)
        SYM(CommInitAll); SYM(CommDestroy); SYM(Reduce); SYM(AllReduce); SYM(GroupStart); SYM(GroupEnd); SYM(GetErrorString);
#undef SYM
        return HPT_OK;
    }
};
Rccl g_rccl;
#define GNCCL(expr)                                                                                                      \
    do {                                                                                                                 \
        ncclResult_t r_ = (expr);                                                                                        \
        if (r_ != ncclSuccess) return hpti_fail(HPT_ERR_DEVICE, "%s failed: %s", #expr, g_rccl.GetErrorString(r_));     \
    } while (0)

struct DevMem {      // grow-only device buffer on one GPU of the group
    void* p = nullptr; size_t bytes = 0;
    int ensure(int device, size_t need) {
        if (need <= bytes && p) return HPT_OK;
        GHIP(hipSetDevice(device));
        if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
        hipError_t e = hipMalloc(&p, need ? need : 16);
        if (e != hipSuccess) { p = nullptr; return hpti_fail(HPT_ERR_NOMEM, "hipMalloc(%zu) on device %d failed: %s", need, device, hipGetErrorString(e)); }
        bytes = need;
        return HPT_OK;
    }
    void release(int device) { if (p) { (void)hipSetDevice(device); (void)hipFree(p); } p = nullptr; bytes = 0; }
};

struct Rank {
    int device = 0;
    hpt_context* ctx = nullptr;
    hpt_scene* replica = nullptr;      // ranks > 0: the group's copy of the caller's scene on this GPU
    DevMem accum, minmax, spp, list[2], stage;
    ncclComm_t comm = nullptr;
};
}  // namespace

struct hpt_group {
    std::vector<Rank> ranks;
    bool rccl = false;                 // transport: RCCL collectives, else peer copies + adds in rank order
    const hpt_scene* source = nullptr; uint64_t generation = 0;   // what the replicas were cloned from
};

// Contiguous, balanced split of `count` samples over the ranks (hpt/dist.py sample_range): rank r gets `n` samples from `lo`
static void sample_range(int rank, int world, int count, int* lo, int* n) {
    const int base = count / world, rem = count % world;
    *lo = rank * base + std::min(rank, rem);
    *n = base + (rank < rem ? 1 : 0);
}

// runs f(rank) on one host thread per GPU (rank 0 on the calling thread); first failure wins, its text becomes this thread's error
template <typename F>
static int per_rank(hpt_group* g, F f) {
    const int n = (int)g->ranks.size();
    std::vector<int> rc((size_t)n, HPT_OK); std::vector<std::string> msg((size_t)n);
    std::vector<std::thread> th;
    auto body = [&](int r) {
        if (hipSetDevice(g->ranks[r].device) != hipSuccess) { rc[r] = HPT_ERR_DEVICE; msg[r] = "hipSetDevice failed"; return; }
        rc[r] = f(r);
        if (rc[r] != HPT_OK) msg[r] = hpt_last_error();
    };
    for (int r = 1; r < n; ++r) th.emplace_back(body, r);
    body(0);
    for (std::thread& t : th) t.join();
    (void)hipSetDevice(g->ranks[0].device);
    for (int r = 0; r < n; ++r) if (rc[r] != HPT_OK) return hpti_fail(rc[r], "GPU %d (rank %d): %s", g->ranks[r].device, r, msg[r].c_str());
    return HPT_OK;
}

extern "C" int hpt_group_create(const int* devices, int n_devices, hpt_group** out) {
    if (!out) return hpti_fail(HPT_ERR_INVALID, "hpt_group_create: out is null");
    *out = nullptr;
    if (n_devices < 0 || n_devices > 64 || (n_devices > 0 && !devices)) return hpti_fail(HPT_ERR_INVALID, "hpt_group_create: bad device list");
    std::vector<int> devs;
    if (n_devices == 0) {      // every GPU the process sees
        const int n = hpt_device_count();
        if (n <= 0) return hpti_fail(HPT_ERR_DEVICE, "no HIP device available: the render core has no CPU fallback");
        for (int i = 0; i < n; ++i) devs.push_back(i);
    } else devs.assign(devices, devices + n_devices);
    hpt_group* g = new hpt_group();
    g->ranks.resize(devs.size());
    bool duplicate = false;
    for (size_t r = 0; r < devs.size(); ++r) {
        g->ranks[r].device = devs[r];
        for (size_t q = 0; q < r; ++q) duplicate = duplicate || devs[q] == devs[r];
        const int rc = hpt_context_create(devs[r], &g->ranks[r].ctx);
        if (rc != HPT_OK) { const std::string keep = hpt_last_error(); hpt_group_destroy(g); return hpti_fail(rc, "%s", keep.c_str()); }
    }
    // transport: RCCL unless a device is listed twice (RCCL refuses that: a rehearsal set-up -- several ranks on one GPU)
    // or HPT_GROUP_TRANSPORT=p2p asks for peer copies + ordered adds
    const char* env = getenv("HPT_GROUP_TRANSPORT");
    g->rccl = !duplicate && !(env && std::strcmp(env, "p2p") == 0);
    if (env && std::strcmp(env, "rccl") == 0 && duplicate) { hpt_group_destroy(g); return hpti_fail(HPT_ERR_INVALID, "hpt_group_create: RCCL cannot put two ranks on one device"); }
    if (g->rccl) {
        int rc = g_rccl.load();
        if (rc == HPT_OK) {
            std::vector<ncclComm_t> comms(devs.size(), nullptr);
            const ncclResult_t r = g_rccl.CommInitAll(comms.data(), (int)devs.size(), devs.data());
            if (r != ncclSuccess) rc = hpti_fail(HPT_ERR_DEVICE, "ncclCommInitAll over %zu GPUs failed: %s", devs.size(), g_rccl.GetErrorString(r));
            else for (size_t k = 0; k < devs.size(); ++k) g->ranks[k].comm = comms[k];
        }
        if (rc != HPT_OK) { const std::string keep = hpt_last_error(); hpt_group_destroy(g); return hpti_fail(rc, "%s", keep.c_str()); }
    } else {
        for (size_t r = 1; r < devs.size(); ++r) {      // peer access GPU 0 <- GPU r for the gather (not needed on one device)
            if (devs[r] == devs[0]) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, devs[0], devs[r]) == hipSuccess && can) {
                (void)hipSetDevice(devs[0]);
                const hipError_t e = hipDeviceEnablePeerAccess(devs[r], 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();      // hipMemcpyPeer stages through the host then
                else (void)hipGetLastError();
            }
        }
    }
    (void)hipSetDevice(devs[0]);
    *out = g;
    return HPT_OK;
}

extern "C" void hpt_group_destroy(hpt_group* g) {
    if (!g) return;
    for (Rank& k : g->ranks) {
        if (k.comm && g_rccl.CommDestroy) { (void)hipSetDevice(k.device); (void)g_rccl.CommDestroy(k.comm); }
        if (k.replica) hpt_scene_destroy(k.replica);
        k.accum.release(k.device); k.minmax.release(k.device); k.spp.release(k.device); k.list[0].release(k.device); k.list[1].release(k.device);
        k.stage.release(k.device);
        if (k.ctx) hpt_context_destroy(k.ctx);      // (deferred by the library while the caller's scene on rank 0 is still alive)
    }
    delete g;
}

extern "C" int hpt_group_size(const hpt_group* g) { return g ? (int)g->ranks.size() : 0; }
extern "C" hpt_context* hpt_group_context(hpt_group* g, int rank) {
    if (!g || rank < 0 || rank >= (int)g->ranks.size()) { (void)hpti_fail(HPT_ERR_INVALID, "hpt_group_context: rank out of range"); return nullptr; }
    return g->ranks[rank].ctx;
}

// the caller's scene lives on rank 0's context; ranks > 0 get clones (again after every re-commit of the source)
static int ensure_replicas(hpt_group* g, hpt_scene* scene) {
    if (hpti_scene_context(scene) != g->ranks[0].ctx) return hpti_fail(HPT_ERR_INVALID, "hpt_group_render: the scene must be created on hpt_group_context(group, 0)");
    const uint64_t gen = hpti_scene_generation(scene);
    if (g->source == scene && g->generation == gen) return HPT_OK;
    for (Rank& k : g->ranks) if (k.replica) { hpt_scene_destroy(k.replica); k.replica = nullptr; }
    g->source = nullptr;
    GCHK(per_rank(g, [&](int r) -> int { return r == 0 ? HPT_OK : hpti_scene_clone(scene, g->ranks[r].ctx, &g->ranks[r].replica); }));
    g->source = scene; g->generation = gen;
    return HPT_OK;
}
static hpt_scene* scene_of(hpt_group* g, hpt_scene* scene, int r) { return r == 0 ? scene : g->ranks[r].replica; }

// sum of the ranks' accumulators onto rank 0 (n int64 each); every rank's stream is drained on return
static int reduce_sum(hpt_group* g, size_t n) {
    const int world = (int)g->ranks.size();
    if (g->rccl) {
        GNCCL(g_rccl.GroupStart());
        for (int r = 0; r < world; ++r) {
            Rank& k = g->ranks[r];
            const ncclResult_t e = g_rccl.Reduce(k.accum.p, k.accum.p, n, ncclInt64, ncclSum, 0, k.comm, hpti_context_stream(k.ctx));
            if (e != ncclSuccess) { (void)g_rccl.GroupEnd(); return hpti_fail(HPT_ERR_DEVICE, "ncclReduce failed: %s", g_rccl.GetErrorString(e)); }
        }
        GNCCL(g_rccl.GroupEnd());
        for (int r = 0; r < world; ++r) { GHIP(hipSetDevice(g->ranks[r].device)); GHIP(hipStreamSynchronize(hpti_context_stream(g->ranks[r].ctx))); }
        GHIP(hipSetDevice(g->ranks[0].device));
        return HPT_OK;
    }
    Rank& root = g->ranks[0];
    GHIP(hipSetDevice(root.device));
    hipStream_t st = hpti_context_stream(root.ctx);
    if (world > 1) GCHK(root.stage.ensure(root.device, n * sizeof(long long)));
    for (int r = 1; r < world; ++r) {
        Rank& k = g->ranks[r];
        if (k.device == root.device) GHIP(hipMemcpyAsync(root.stage.p, k.accum.p, n * sizeof(long long), hipMemcpyDeviceToDevice, st));
        else GHIP(hipMemcpyPeerAsync(root.stage.p, root.device, k.accum.p, k.device, n * sizeof(long long), st));
        hptk_add_i64(st, hpti_context_max_blocks(root.ctx), (long long*)root.accum.p, (const long long*)root.stage.p, n);
        GHIP(hipGetLastError());
    }
    GHIP(hipStreamSynchronize(st));
    return HPT_OK;
}

// per-channel min/max planes (npix x 6) of all ranks folded together, the result on every rank
static int allreduce_minmax(hpt_group* g, size_t n) {
    const int world = (int)g->ranks.size();
    if (world == 1) return HPT_OK;
    if (g->rccl) {
        for (int r = 0; r < world; ++r) {
            Rank& k = g->ranks[r];
            GHIP(hipSetDevice(k.device));
            hptk_negate_min(hpti_context_stream(k.ctx), hpti_context_max_blocks(k.ctx), (float*)k.minmax.p, n);
        }
        GNCCL(g_rccl.GroupStart());
        for (int r = 0; r < world; ++r) {
            Rank& k = g->ranks[r];
            const ncclResult_t e = g_rccl.AllReduce(k.minmax.p, k.minmax.p, n, ncclFloat, ncclMax, k.comm, hpti_context_stream(k.ctx));
            if (e != ncclSuccess) { (void)g_rccl.GroupEnd(); return hpti_fail(HPT_ERR_DEVICE, "ncclAllReduce failed: %s", g_rccl.GetErrorString(e)); }
        }
        GNCCL(g_rccl.GroupEnd());
        for (int r = 0; r < world; ++r) {
            Rank& k = g->ranks[r];
            GHIP(hipSetDevice(k.device));
            hptk_negate_min(hpti_context_stream(k.ctx), hpti_context_max_blocks(k.ctx), (float*)k.minmax.p, n);
            GHIP(hipGetLastError());
            GHIP(hipStreamSynchronize(hpti_context_stream(k.ctx)));
        }
        GHIP(hipSetDevice(g->ranks[0].device));
        return HPT_OK;
    }
    Rank& root = g->ranks[0];
    GHIP(hipSetDevice(root.device));
    hipStream_t st = hpti_context_stream(root.ctx);
    GCHK(root.stage.ensure(root.device, n * sizeof(float)));
    for (int r = 1; r < world; ++r) {
        Rank& k = g->ranks[r];
        if (k.device == root.device) GHIP(hipMemcpyAsync(root.stage.p, k.minmax.p, n * sizeof(float), hipMemcpyDeviceToDevice, st));
        else GHIP(hipMemcpyPeerAsync(root.stage.p, root.device, k.minmax.p, k.device, n * sizeof(float), st));
        hptk_minmax_merge(st, hpti_context_max_blocks(root.ctx), (float*)root.minmax.p, (const float*)root.stage.p, n);
        GHIP(hipGetLastError());
    }
    for (int r = 1; r < world; ++r) {
        Rank& k = g->ranks[r];
        if (k.device == root.device) GHIP(hipMemcpyAsync(k.minmax.p, root.minmax.p, n * sizeof(float), hipMemcpyDeviceToDevice, st));
        else GHIP(hipMemcpyPeerAsync(k.minmax.p, k.device, root.minmax.p, root.device, n * sizeof(float), st));
    }
    GHIP(hipStreamSynchronize(st));
    return HPT_OK;
}

extern "C" int hpt_group_render(hpt_group* g, hpt_scene* scene, const hpt_render_params* p, float* rgb, uint32_t* samples_per_px,
                                hpt_group_stats* stats) {
    if (!g || !scene || !p || !rgb) return hpti_fail(HPT_ERR_INVALID, "hpt_group_render: null argument");
    if (p->adaptive ? (p->initial_samples <= 0 || p->max_samples < p->initial_samples) : (p->spp <= 0 || p->first_sample < 0))
        return hpti_fail(HPT_ERR_INVALID, p->adaptive ? "adaptive: need 0 < initial_samples <= max_samples" : "need spp > 0 and first_sample >= 0");
    const double t0 = now_s();
    const int world = (int)g->ranks.size();
    GCHK(ensure_replicas(g, scene));
    uint32_t w = 0, h = 0;
    hpti_scene_frame(scene, &w, &h);
    const size_t npix = (size_t)w * h;
    if (stats) { std::memset(stats, 0, sizeof *stats); stats->n_devices = world; stats->transport = g->rccl ? 0 : 1; }
    std::vector<hpt_render_stats> rst((size_t)world);
    std::vector<double> busy((size_t)world, 0.0);
    for (Rank& k : g->ranks) GCHK(k.accum.ensure(k.device, npix * sizeof(hpt_accum)));
    double reduce_s = 0.0;
    Rank& root = g->ranks[0];
    hipStream_t st0 = hpti_context_stream(root.ctx);
    if (!p->adaptive) {
        // ---- fixed spp: rank r renders samples [first + lo_r, first + lo_r + n_r) of every pixel
        GCHK(per_rank(g, [&](int r) -> int {
            Rank& k = g->ranks[r];
            int lo, n; sample_range(r, world, p->spp, &lo, &n);
            const double t = now_s();
            GHIP(hipMemsetAsync(k.accum.p, 0, npix * sizeof(hpt_accum), hpti_context_stream(k.ctx)));
            if (n == 0) {
                GHIP(hipStreamSynchronize(hpti_context_stream(k.ctx)));
                return HPT_OK;
            }
            hpt_render_params q = *p; q.first_sample = p->first_sample + lo; q.spp = n;
            const int rc = hpt_render_accum_device(scene_of(g, scene, r), &q, nullptr, 0, (hpt_accum*)k.accum.p, nullptr, &rst[r]);      // (stats given: synchronises)
            busy[r] = now_s() - t;
            return rc;
        }));
        const double tr = now_s();
        GCHK(reduce_sum(g, npix * 4));
        reduce_s += now_s() - tr;
        GHIP(hipSetDevice(root.device));
        GCHK(root.stage.ensure(root.device, npix * 3 * sizeof(float)));
        hptk_accum_resolve(st0, hpti_context_max_blocks(root.ctx), (const long long*)root.accum.p, nullptr, (float*)root.stage.p, (uint32_t)npix, (float)p->spp);      // main.rs:110
        GHIP(hipGetLastError());
        GHIP(hipStreamSynchronize(st0));
        GHIP(hipMemcpy(rgb, root.stage.p, npix * 3 * sizeof(float), hipMemcpyDeviceToHost));
        if (samples_per_px) for (size_t i = 0; i < npix; ++i) samples_per_px[i] = (uint32_t)p->spp;
    } else {
        // ---- main.rs:67-107 over the ranks, rounds in lock step
        for (Rank& k : g->ranks) {
            GCHK(k.minmax.ensure(k.device, npix * 6 * sizeof(float))); GCHK(k.spp.ensure(k.device, npix * sizeof(uint32_t)));
            GCHK(k.list[0].ensure(k.device, npix * sizeof(uint32_t))); GCHK(k.list[1].ensure(k.device, npix * sizeof(uint32_t)));
        }
        GCHK(per_rank(g, [&](int r) -> int {
            Rank& k = g->ranks[r];
            hipStream_t st = hpti_context_stream(k.ctx);
            GHIP(hipMemsetAsync(k.accum.p, 0, npix * sizeof(hpt_accum), st));
            hptk_init_minmax(st, hpti_context_max_blocks(k.ctx), (float*)k.minmax.p, (uint32_t)npix);                       // main.rs:90-92
            hptk_fill_u32(st, hpti_context_max_blocks(k.ctx), (uint32_t*)k.spp.p, npix, (uint32_t)p->initial_samples);
            GHIP(hipGetLastError());
            GHIP(hipStreamSynchronize(st));
            return HPT_OK;
        }));
        int have = 0, num = p->initial_samples, cur = -1;      // cur: which list holds the active pixels (-1: all of them)
        uint32_t n_active = (uint32_t)npix;
        while (n_active > 0) {
            GCHK(per_rank(g, [&](int r) -> int {
                Rank& k = g->ranks[r];
                int lo, n; sample_range(r, world, num - have, &lo, &n);
                if (n == 0) return HPT_OK;
                const double t = now_s();
                hpt_render_params q = *p; q.adaptive = 0; q.first_sample = have + lo; q.spp = n;
                hpt_render_stats one;
                const int rc = hpt_render_accum_device(scene_of(g, scene, r), &q, cur < 0 ? nullptr : (const uint32_t*)k.list[cur].p, n_active,
                                                       (hpt_accum*)k.accum.p, (float*)k.minmax.p, &one);
                busy[r] += now_s() - t;
                rst[r].rays_primary += one.rays_primary; rst[r].rays_bounce += one.rays_bounce; rst[r].samples += one.samples; rst[r].passes += one.passes;
                return rc;
            }));
            const double tr = now_s();
            GCHK(allreduce_minmax(g, npix * 6));
            reduce_s += now_s() - tr;
            have = num;
            if (num >= p->max_samples) break;      // main.rs:101, first clause
            const int next = cur < 0 ? 0 : 1 - cur;
            std::vector<uint32_t> n_out((size_t)world, 0u);
            GCHK(per_rank(g, [&](int r) -> int {
                Rank& k = g->ranks[r];
                return hpt_adaptive_select_device(scene_of(g, scene, r), cur < 0 ? nullptr : (const uint32_t*)k.list[cur].p, n_active, (const float*)k.minmax.p,
                                                  p->thresholds, (uint32_t*)k.list[next].p, (uint32_t*)k.spp.p, (uint32_t)(num * 2), &n_out[r]);
            }));
            for (int r = 1; r < world; ++r)
                if (n_out[r] != n_out[0]) return hpti_fail(HPT_ERR_STATE, "hpt_group_render: ranks disagree on the active set (%u vs %u pixels)", n_out[r], n_out[0]);
            n_active = n_out[0]; cur = next; num *= 2;
        }
        const double tr = now_s();
        GCHK(reduce_sum(g, npix * 4));
        reduce_s += now_s() - tr;
        GHIP(hipSetDevice(root.device));
        GCHK(root.stage.ensure(root.device, npix * 3 * sizeof(float)));
        hptk_accum_resolve(st0, hpti_context_max_blocks(root.ctx), (const long long*)root.accum.p, (const uint32_t*)root.spp.p, (float*)root.stage.p, (uint32_t)npix, 0.0f);
        GHIP(hipGetLastError());
        GHIP(hipStreamSynchronize(st0));
        GHIP(hipMemcpy(rgb, root.stage.p, npix * 3 * sizeof(float), hipMemcpyDeviceToHost));
        if (samples_per_px) GHIP(hipMemcpy(samples_per_px, root.spp.p, npix * sizeof(uint32_t), hipMemcpyDeviceToHost));
    }
    if (stats) {
        for (int r = 0; r < world; ++r) {
            stats->rays_primary += rst[r].rays_primary; stats->rays_bounce += rst[r].rays_bounce; stats->samples += rst[r].samples;
            stats->seconds_render_max = std::max(stats->seconds_render_max, busy[r]);
        }
        stats->seconds_reduce = reduce_s;
        stats->seconds_total = now_s() - t0;
    }
    return HPT_OK;
}
