// hpt_api.cpp -- the C ABI of include/hpt.h: scene hand-over, flatten + upload, and the wavefront render driver.
// Host code only (kernels live in hpt_kernels.hip).  Compiled with -ffp-contract=off: the camera frame and the
// matrix inverses below are computed in the reference's f32 operation order.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/hpt.h"
#include "hpt_bvh.h"
#include "hpt_device.h"
#include "hpt_internal.h"
#include "hpt_launch.h"

// ------------------------------------------------------------------------------------------------ errors
static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_err = buf;
    return code;
}
#define HIPCHK(expr)                                                                                       \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess) return fail(HPT_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define CHK(expr) do { int rc_ = (expr); if (rc_ != HPT_OK) return rc_; } while (0)

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ------------------------------------------------------------------------------------------------ device buffers
struct DevBuf {
    void* p = nullptr; size_t bytes = 0;
    int ensure(size_t need) {   // grow-only workspace
        if (need <= bytes && p) return HPT_OK;
        if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
        if (need == 0) need = 16;
        hipError_t e = hipMalloc(&p, need);
        if (e != hipSuccess) { p = nullptr; return fail(HPT_ERR_NOMEM, "hipMalloc(%zu) failed: %s", need, hipGetErrorString(e)); }
        bytes = need;
        return HPT_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
    template <typename T> T* as() const { return (T*)p; }
};

template <typename T>
static int upload(DevBuf& b, const std::vector<T>& v, hipStream_t st) {
    CHK(b.ensure(v.size() * sizeof(T)));
    if (!v.empty()) HIPCHK(hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, st));
    return HPT_OK;
}

// ------------------------------------------------------------------------------------------------ context / scene
#define HPT_MAX_INFLIGHT 4
struct hpt_context {
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    // two passes in flight (render_range): odd passes run on aux_stream with a workspace of their own, so the ends of one
    // pass's launches (a persistent grid drains unevenly) are filled by the other pass's kernels
    hipStream_t aux_stream[HPT_MAX_INFLIGHT - 1] = {};
    hipEvent_t ev_fork = nullptr, ev_done[HPT_MAX_INFLIGHT] = {};
    int overlap = 2;            // passes in flight (1 = one stream)
    int extend_spare = -1;      // k_extend workgroup slots per CU left to the other pass when two are in flight; -1: whatever leaves a grid of 4 per CU
                                // (6 resident: spare 0 / 1 / 2 / 3 = 13.42 / 13.41 / 14.05 / 13.59 Grays/s; 7 resident: 1 / 2 / 3 = 14.13 / 14.05 / 14.41)
    int num_cus = 256;
    int max_blocks = 2048;
    int extend_blocks = 1024;   // persistent grid of the test-entry traversal kernels
    int extend_per_cu = 0;      // workgroups per CU of k_extend; 0 = by kernel variant and LDS need (extend_grid)
    std::vector<hipEvent_t> events;   // pool for HPT_RENDER_TIME_KERNELS and the trace entry's kernel timing
    int live_scenes = 0;              // scenes created on this context and not yet destroyed
    bool destroy_pending = false;     // hpt_context_destroy was called while scenes were live: freed with the last scene
    void* scratch = nullptr; size_t scratch_bytes = 0;   // this device's copy of the static no-alloc workspace (hptk_scratch)
};

struct HostMesh {
    std::vector<float> verts, normals, uvw;
    std::vector<uint32_t> tri, ntri, ttri;
    int32_t n_tri = 0;
    HostTree tree;
    bool host_tree = false;   // tree handed over by the host (bvh.rs:35-146 output) rather than built here
};
struct HostNode { int32_t parent, geom, mesh, material; float m[9], t[3]; std::vector<int32_t> children; };
struct HostImage { std::vector<uint8_t> px; int32_t w, h; };

struct hpt_scene {
    hpt_context* ctx = nullptr;
    std::vector<HostMesh> meshes;
    std::vector<HostNode> nodes;
    std::vector<int32_t> top;
    std::vector<hpt_material> materials;
    std::vector<HostImage> images;
    hpt_texture background, environment;
    hpt_camera camera;
    bool have_camera = false, committed = false;
    uint64_t generation = 0;    // stamp of the last commit, unique in the process (a group re-replicates its clones when the stamp of
                                // the scene it is handed differs -- a re-commit, or ANOTHER scene that the allocator put at the same address)
    // device side
    DevBuf d_nodes, d_tris, d_trishade, d_meshes, d_instances, d_meshinst, d_xinv, d_xfwd, d_materials, d_images, d_pool, d_pnormal;
    std::vector<DMesh> h_meshes;
    DScene ds;
    int stack_entries = 1;
    int n_xf = 0, n_mat = 0;
    HptkPlan plan;              // launch configuration of this scene's kernels (hpt_scene_commit)
    int shade_grid = 0;         // k_shade workgroups per launch
    bool fuse_analytic = false; // k_shade of depth d runs the analytic stage of depth d + 1 (instance-list path; set at commit)
    int analytic_grid = 0;      // k_extend_a workgroups per launch
    int32_t tlas_nodes = 0, node_bytes = 64, trees_optimized = 0; int64_t n_tris_total = 0; uint64_t device_bytes = 0;
    // render workspace (grow-only, reused across calls: nothing is allocated in a steady-state render call)
    DevBuf w_q[2][3], w_hit, w_hit_inst, w_mq, w_mq_bin, w_radiance, w_counts, w_halton, w_stat, w_accum, w_minmax, w_spp, w_list[2], w_listcount, w_out;
    struct PassBufs { DevBuf q[2][3], hit, hit_inst, mq, mq_bin, radiance, counts; } x[HPT_MAX_INFLIGHT - 1];   // pass buffers of the further passes in flight
    // trace workspace
    DevBuf t_org, t_dir, t_hit, t_side, t_misc[9], t_mq, t_mqcount;
};

static void identity_tex(hpt_texture& t) {   // load.rs:30-34
    std::memset(&t, 0, sizeof t);
    t.m[0] = t.m[4] = t.m[8] = 1.0f;
}

// cgmath Matrix3::invert (column-major): from_cols(c1 x c2 / det, c2 x c0 / det, c0 x c1 / det).transpose()
static bool mat3_invert(const float* m, float* out) {
#define A(c, r) m[(c) * 3 + (r)]
    const float det = A(0, 0) * (A(1, 1) * A(2, 2) - A(2, 1) * A(1, 2)) - A(1, 0) * (A(0, 1) * A(2, 2) - A(2, 1) * A(0, 2)) +
                      A(2, 0) * (A(0, 1) * A(1, 2) - A(1, 1) * A(0, 2));
    if (det == 0.0f || det != det) return false;
    const float c0[3] = {A(0, 0), A(0, 1), A(0, 2)}, c1[3] = {A(1, 0), A(1, 1), A(1, 2)}, c2[3] = {A(2, 0), A(2, 1), A(2, 2)};
#undef A
    auto cross = [](const float* a, const float* b, float* r) {
        r[0] = a[1] * b[2] - a[2] * b[1]; r[1] = a[2] * b[0] - a[0] * b[2]; r[2] = a[0] * b[1] - a[1] * b[0];
    };
    float r0[3], r1[3], r2[3];
    cross(c1, c2, r0); cross(c2, c0, r1); cross(c0, c1, r2);
    for (int k = 0; k < 3; ++k) { r0[k] = r0[k] / det; r1[k] = r1[k] / det; r2[k] = r2[k] / det; }
    // rows of the result: out(c, r=0) = r0[c] ...
    for (int c = 0; c < 3; ++c) { out[c * 3 + 0] = r0[c]; out[c * 3 + 1] = r1[c]; out[c * 3 + 2] = r2[c]; }
    return true;
}

// ------------------------------------------------------------------------------------------------ misc exports
extern "C" const char* hpt_version(void) { return "hpt 0.1 (gfx950)"; }
extern "C" const char* hpt_last_error(void) { return g_err.c_str(); }
extern "C" int hpt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
static inline uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
extern "C" float hpt_rng_f32(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t dim) {
    const uint32_t k1 = mix32(pixel + mix32(seed));
    const uint32_t k2 = mix32(sample ^ 0x5bd1e995U ^ (seed * 0x9E3779B9U));
    const uint32_t h = mix32(k1 ^ (k2 + dim * 0x9E3779B9U));
    return (float)(h >> 8) * (1.0f / 16777216.0f);
}
extern "C" float hpt_halton(int32_t index, int32_t base) {   // scene.rs:431-443
    float r = 0.0f, f = 1.0f;
    for (int32_t i = index; i > 0; i /= base) {
        f /= (float)base;
        r += f * (float)(i % base);
    }
    return r;
}
extern "C" int hpt_tonemap_rgba8(const float* rgb, uint32_t n, uint8_t* rgba) {   // scene.rs:360-367
    if (!rgb || !rgba) return fail(HPT_ERR_INVALID, "hpt_tonemap_rgba8: null pointer");
    const float GAMMA = 1.0f / 2.2f;
    for (uint32_t i = 0; i < n; ++i) {
        for (int c = 0; c < 3; ++c) {
            float v = powf(rgb[3 * (size_t)i + c], GAMMA) * 255.0f;
            v = fminf(fmaxf(v, 0.0f), 255.0f);   // Rust max/min drop NaN -> 0
            rgba[4 * (size_t)i + c] = (uint8_t)v;
        }
        rgba[4 * (size_t)i + 3] = 255;
    }
    return HPT_OK;
}

static int context_init(hpt_context* c);
static void context_free(hpt_context* c);
extern "C" int hpt_context_create(int device, hpt_context** out) {
    if (!out) return fail(HPT_ERR_INVALID, "hpt_context_create: out is null");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(HPT_ERR_DEVICE, "no HIP device available (%s): the render core has no CPU fallback", hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(HPT_ERR_INVALID, "device %d out of range (0..%d)", device, n - 1);
    HIPCHK(hipSetDevice(device));
    hpt_context* c = new hpt_context();
    c->device = device;
    const int rc = context_init(c);
    if (rc != HPT_OK) { context_free(c); return rc; }      // (streams and events created so far go with it)
    *out = c;
    return HPT_OK;
}
static int context_init(hpt_context* c) {
    const int device = c->device;
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    c->max_blocks = c->num_cus * 8;
    c->extend_blocks = c->num_cus * 4;
    if (const char* e = getenv("HPT_EXTEND_BLOCKS_PER_CU")) { int v = atoi(e); if (v >= 1 && v <= 16) c->extend_per_cu = v; }
    if (const char* e = getenv("HPT_EXTEND_SPARE")) c->extend_spare = std::min(std::max(atoi(e), 0), 3);
    if (const char* e = getenv("HPT_PASSES_IN_FLIGHT")) c->overlap = std::min(std::max(atoi(e), 1), HPT_MAX_INFLIGHT);
    HIPCHK(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    for (hipStream_t& a : c->aux_stream) HIPCHK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    for (hipEvent_t& e : c->ev_done) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    c->stream = c->own_stream;
    // this device's address of the static workspace of the no-alloc entry point (once per device, with it current)
    // (a failure leaves it null: only hpt_render_device_noalloc needs it, and reports HPT_ERR_STATE then)
    if (hptk_scratch(device, &c->scratch, &c->scratch_bytes) != 0) { c->scratch = nullptr; c->scratch_bytes = 0; }
    return HPT_OK;
}
// Scenes keep a pointer to their context (device, streams): destroying the context first would leave them dangling
// -- Python's __del__ order at interpreter exit does exactly that.  So a context with live scenes is only marked; the
// last hpt_scene_destroy frees it.
extern "C" void hpt_context_destroy(hpt_context* c) {
    if (!c) return;
    if (c->live_scenes > 0) { c->destroy_pending = true; return; }
    context_free(c);
}
static void context_free(hpt_context* c) {
    (void)hipSetDevice(c->device);
    for (hipEvent_t e : c->events) (void)hipEventDestroy(e);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    for (hipEvent_t e : c->ev_done) if (e) (void)hipEventDestroy(e);
    for (hipStream_t a : c->aux_stream) if (a) (void)hipStreamDestroy(a);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}
extern "C" int hpt_context_set_stream(hpt_context* c, void* s) {
    if (!c) return fail(HPT_ERR_INVALID, "null context");
    c->stream = s ? (hipStream_t)s : c->own_stream;
    return HPT_OK;
}
extern "C" int hpt_context_set_passes_in_flight(hpt_context* c, int n) {
    if (!c) return fail(HPT_ERR_INVALID, "null context");
    if (n < 1 || n > HPT_MAX_INFLIGHT) return fail(HPT_ERR_INVALID, "passes in flight must be in 1..%d", HPT_MAX_INFLIGHT);
    c->overlap = n;
    return HPT_OK;
}
extern "C" int hpt_context_synchronize(hpt_context* c) {
    if (!c) return fail(HPT_ERR_INVALID, "null context");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    return HPT_OK;
}

extern "C" int hpt_scene_create(hpt_context* ctx, hpt_scene** out) {
    if (!ctx || !out) return fail(HPT_ERR_INVALID, "hpt_scene_create: null argument");
    if (ctx->destroy_pending) return fail(HPT_ERR_STATE, "hpt_scene_create: the context has been destroyed");
    hpt_scene* s = new hpt_scene();
    s->ctx = ctx; ctx->live_scenes++;
    identity_tex(s->background); identity_tex(s->environment);
    const hpt_camera cam = {{0, 0, 0}, {0, 0, -1}, {0, 1, 0}, 40.0f, 800, 600, 1.0f, 0.0f};   // scene.rs:107-120
    s->camera = cam;
    std::memset(&s->ds, 0, sizeof s->ds);
    *out = s;
    return HPT_OK;
}
extern "C" void hpt_scene_destroy(hpt_scene* s) {
    if (!s) return;
    (void)hipSetDevice(s->ctx->device);
    (void)hipStreamSynchronize(s->ctx->stream);
    for (hipStream_t a : s->ctx->aux_stream) if (a) (void)hipStreamSynchronize(a);   // (joined already unless a render failed half way)
    DevBuf* all[] = {&s->d_nodes, &s->d_tris, &s->d_trishade, &s->d_meshes, &s->d_instances, &s->d_meshinst, &s->d_xinv, &s->d_xfwd, &s->d_pnormal, &s->d_materials,
                     &s->d_images, &s->d_pool, &s->w_hit, &s->w_hit_inst, &s->w_mq, &s->w_mq_bin, &s->t_mq, &s->t_mqcount, &s->w_radiance, &s->w_counts, &s->w_halton, &s->w_stat,
                     &s->w_accum, &s->w_minmax, &s->w_spp, &s->w_list[0], &s->w_list[1], &s->w_listcount, &s->w_out,
                     &s->t_org, &s->t_dir, &s->t_hit, &s->t_side};
    for (DevBuf* b : all) b->release();
    for (int a = 0; a < 2; ++a) for (int k = 0; k < 3; ++k) s->w_q[a][k].release();
    for (auto& x : s->x) {
        x.hit.release(); x.hit_inst.release(); x.mq.release(); x.mq_bin.release(); x.radiance.release(); x.counts.release();
        for (int a = 0; a < 2; ++a) for (int k = 0; k < 3; ++k) x.q[a][k].release();
    }
    for (DevBuf& b : s->t_misc) b.release();
    hpt_context* ctx = s->ctx;
    delete s;
    if (--ctx->live_scenes == 0 && ctx->destroy_pending) context_free(ctx);
}

extern "C" int hpt_scene_add_mesh(hpt_scene* s, const float* verts, int32_t nv, const uint32_t* tri_idx,
                                  const float* normals, int32_t nn, const uint32_t* nrm_idx,
                                  const float* uvw, int32_t nt, const uint32_t* uv_idx, int32_t ntri,
                                  const hpt_bvh_node* nodes, int32_t nnodes, int32_t* mesh_id) {
    if (!s || !verts || !tri_idx || !normals || !nrm_idx || !uvw || !uv_idx) return fail(HPT_ERR_INVALID, "hpt_scene_add_mesh: null pointer (vt and vn are mandatory, load.rs:303-304)");
    if (nv <= 0 || nn <= 0 || nt <= 0) return fail(HPT_ERR_INVALID, "hpt_scene_add_mesh: empty vertex/normal/texture array");
    if (ntri <= 0) return fail(HPT_ERR_INVALID, "Cannot construct empty BVH");   // bvh.rs:37
    if (ntri > (1 << 28)) return fail(HPT_ERR_LIMIT, "hpt_scene_add_mesh: more than 2^28 triangles in one mesh");
    for (int64_t i = 0; i < (int64_t)ntri * 3; ++i) {
        if (tri_idx[i] >= (uint32_t)nv) return fail(HPT_ERR_INVALID, "hpt_scene_add_mesh: vertex index %u out of range at triangle %lld", tri_idx[i], (long long)(i / 3));
        if (nrm_idx[i] >= (uint32_t)nn) return fail(HPT_ERR_INVALID, "hpt_scene_add_mesh: normal index %u out of range at triangle %lld", nrm_idx[i], (long long)(i / 3));
        if (uv_idx[i] >= (uint32_t)nt) return fail(HPT_ERR_INVALID, "hpt_scene_add_mesh: texture index %u out of range at triangle %lld", uv_idx[i], (long long)(i / 3));
    }
    HostMesh m;
    m.verts.assign(verts, verts + (size_t)nv * 3);
    m.normals.assign(normals, normals + (size_t)nn * 3);
    m.uvw.assign(uvw, uvw + (size_t)nt * 3);
    m.tri.assign(tri_idx, tri_idx + (size_t)ntri * 3);
    m.ntri.assign(nrm_idx, nrm_idx + (size_t)ntri * 3);
    m.ttri.assign(uv_idx, uv_idx + (size_t)ntri * 3);
    m.n_tri = ntri;
    if (nodes) {
        std::string err;
        if (!tree_from_reference_nodes(nodes, nnodes, ntri, m.tree, err)) return fail(HPT_ERR_INVALID, "%s", err.c_str());
        m.host_tree = true;
    } else {
        tree_build_sah(m.verts.data(), m.tri.data(), ntri, m.tree);
    }
    s->meshes.push_back(std::move(m));
    s->committed = false;
    if (mesh_id) *mesh_id = (int32_t)s->meshes.size() - 1;
    return HPT_OK;
}
extern "C" int hpt_scene_add_image(hpt_scene* s, const uint8_t* rgb8, int32_t w, int32_t h, int32_t* image_id) {
    if (!s || w < 0 || h < 0 || (!rgb8 && (int64_t)w * h > 0)) return fail(HPT_ERR_INVALID, "hpt_scene_add_image: bad argument");
    // scene.rs:330 indexes x % width, y % height: an image with exactly one zero dimension panics there (division by zero);
    // 0 x 0 is the reference's "no data" image (scene.rs:323 returns black)
    if ((w == 0) != (h == 0)) return fail(HPT_ERR_INVALID, "hpt_scene_add_image: %d x %d image (one dimension is zero)", w, h);
    if ((int64_t)w * h > (1ll << 28)) return fail(HPT_ERR_LIMIT, "hpt_scene_add_image: more than 2^28 pixels");
    HostImage im; im.w = w; im.h = h;
    im.px.assign(rgb8, rgb8 + (size_t)w * (size_t)h * 3);
    s->images.push_back(std::move(im));
    s->committed = false;
    if (image_id) *image_id = (int32_t)s->images.size() - 1;
    return HPT_OK;
}
static int check_texture(const hpt_scene* s, const hpt_texture& t, const char* what) {
    if (t.kind < HPT_TEX_BLANK || t.kind > HPT_TEX_IMAGE) return fail(HPT_ERR_INVALID, "%s: unknown texture kind %d", what, t.kind);
    if (t.kind == HPT_TEX_IMAGE && (t.image < 0 || t.image >= (int32_t)s->images.size())) return fail(HPT_ERR_INVALID, "%s: image id %d does not exist", what, t.image);
    float inv[9];
    if (!mat3_invert(t.m, inv)) return fail(HPT_ERR_INVALID, "%s: texture transform is singular (scene.rs:279 unwrap)", what);
    return HPT_OK;
}
extern "C" int hpt_scene_add_material(hpt_scene* s, const hpt_material* m, int32_t* id) {
    if (!s || !m) return fail(HPT_ERR_INVALID, "hpt_scene_add_material: null argument");
    CHK(check_texture(s, m->diffuse, "material.diffuse"));
    CHK(check_texture(s, m->reflection, "material.reflection"));
    CHK(check_texture(s, m->refraction, "material.refraction"));
    s->materials.push_back(*m);
    s->committed = false;
    if (id) *id = (int32_t)s->materials.size() - 1;
    return HPT_OK;
}
extern "C" int hpt_scene_add_node(hpt_scene* s, int32_t parent, int32_t geom, int32_t mesh, int32_t material,
                                  const float m[9], const float t[3], int32_t* node_id) {
    if (!s || !m || !t) return fail(HPT_ERR_INVALID, "hpt_scene_add_node: null argument");
    if (parent < -1 || parent >= (int32_t)s->nodes.size()) return fail(HPT_ERR_INVALID, "hpt_scene_add_node: parent %d does not exist", parent);
    if (geom < HPT_GEOM_NONE || geom > HPT_GEOM_MESH) return fail(HPT_ERR_INVALID, "unknown object type");   // load.rs:92
    if (geom == HPT_GEOM_MESH && (mesh < 0 || mesh >= (int32_t)s->meshes.size())) return fail(HPT_ERR_INVALID, "hpt_scene_add_node: mesh id %d does not exist", mesh);
    if (geom != HPT_GEOM_NONE && (material < 0 || material >= (int32_t)s->materials.size())) return fail(HPT_ERR_INVALID, "material does not exist for object");   // scene.rs:133
    float inv[9];
    if (!mat3_invert(m, inv)) return fail(HPT_ERR_INVALID, "hpt_scene_add_node: node transform is singular (scene.rs:279 unwrap)");
    HostNode n;
    n.parent = parent; n.geom = geom; n.mesh = mesh; n.material = material;
    std::memcpy(n.m, m, sizeof n.m); std::memcpy(n.t, t, sizeof n.t);
    s->nodes.push_back(n);
    const int32_t id = (int32_t)s->nodes.size() - 1;
    if (parent < 0) s->top.push_back(id); else s->nodes[parent].children.push_back(id);
    s->committed = false;
    if (node_id) *node_id = id;
    return HPT_OK;
}
extern "C" int hpt_scene_set_background(hpt_scene* s, const hpt_texture* t) {
    if (!s || !t) return fail(HPT_ERR_INVALID, "null argument");
    CHK(check_texture(s, *t, "background"));
    s->background = *t; s->committed = false;
    return HPT_OK;
}
extern "C" int hpt_scene_set_environment(hpt_scene* s, const hpt_texture* t) {
    if (!s || !t) return fail(HPT_ERR_INVALID, "null argument");
    CHK(check_texture(s, *t, "environment"));
    s->environment = *t; s->committed = false;
    return HPT_OK;
}
extern "C" int hpt_scene_set_camera(hpt_scene* s, const hpt_camera* c) {
    if (!s || !c) return fail(HPT_ERR_INVALID, "null argument");
    if (c->img_width == 0 || c->img_height == 0) return fail(HPT_ERR_INVALID, "camera: zero image size");
    if ((uint64_t)c->img_width * c->img_height > (1ull << 26)) return fail(HPT_ERR_LIMIT, "camera: more than 2^26 pixels");
    s->camera = *c; s->have_camera = true; s->committed = false;
    return HPT_OK;
}

static void conv_texture(const hpt_texture& t, DTexture& d) {
    d.kind = t.kind; d.image = t.image;
    for (int k = 0; k < 3; ++k) { d.color[k] = t.color[k]; d.color1[k] = t.color1[k]; d.color2[k] = t.color2[k]; d.t[k] = t.t[k]; }
    float inv[9];
    mat3_invert(t.m, inv);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) d.minv[r * 3 + c] = inv[c * 3 + r];
}

// Tree restructuring at flatten time (csrc/hpt_bvh.cpp tree_optimize).  HPT_TREE_OPT=0 keeps the handed-over topology
// (tight refit only, the round-1 behaviour; A/B runs and the invariance tests), HPT_TREE_OPT_PASSES = insertion rounds.
// Depth: the per-lane traversal stack lives in LDS, one KB per entry and workgroup; up to 17 interior levels the
// instance-list kernel keeps its 5 workgroups per CU (18 entries + 12.8 KB of rings and lists = 31.5 KB; the CU hands
// LDS out in 1 280-B granules, 128 of them, so 25 per workgroup: measured -- with 19 entries the occupancy query still
// answered 5, four were resident and the launch took 35 % longer), so the insertion rounds may not push a tree beyond
// that (or beyond its own depth, if it came deeper).
#define HPT_SOFT_TREE_DEPTH 17
#define HPT_HARD_TREE_DEPTH 38
struct TreeOpt { int mode, passes; int32_t max_leaves_for_passes; };
static int g_tree_opt_override = -1;
static TreeOpt tree_opt_settings() {
    TreeOpt o{1, 4, 1 << 18};
    if (const char* e = getenv("HPT_TREE_OPT")) o.mode = atoi(e) != 0;
    if (const char* e = getenv("HPT_TREE_OPT_PASSES")) o.passes = std::min(std::max(atoi(e), 0), 64);
    if (const char* e = getenv("HPT_TREE_OPT_MAX_LEAVES")) o.max_leaves_for_passes = std::max(atoi(e), 0);
    if (g_tree_opt_override >= 0) o.mode = g_tree_opt_override;
    return o;
}

// main.rs:43-53
static void camera_frame(const hpt_camera& c, DCamera& f) {
    const float PI = 3.14159265358979323846f;
    const float height = tanf((c.fov / 2.0f) * (2.0f * PI / 360.0f)) * 2.0f * c.focaldist;
    const float width = height * (float)c.img_width / (float)c.img_height;
    f.pixel_height = height / (float)c.img_height;
    f.pixel_width = width / (float)c.img_width;
    const float* d = c.dir; const float* u = c.up;
    const float right[3] = {d[1] * u[2] - d[2] * u[1], d[2] * u[0] - d[0] * u[2], d[0] * u[1] - d[1] * u[0]};
    for (int k = 0; k < 3; ++k) {
        const float b = (c.pos[k] + d[k] * c.focaldist) + u[k] * (height / 2.0f);
        f.a[k] = b - right[k] * (width / 2.0f);
        f.right[k] = right[k]; f.up[k] = u[k]; f.pos[k] = c.pos[k];
    }
    f.dof = c.dof; f.w = c.img_width; f.h = c.img_height;
}

extern "C" int hpt_scene_commit(hpt_scene* s) {
    if (!s) return fail(HPT_ERR_INVALID, "null scene");
    hpt_context* ctx = s->ctx;
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    // ---- meshes: flatten and concatenate (child refs rebased to global node / triangle indices)
    std::vector<float> nodes, tris, trishade;
    struct TreeSlice { int64_t node_base; int32_t n_interior; QuantFrame qf; };
    std::vector<TreeSlice> trees;   // every tree in nodes[] (one per mesh, then the TLAS) with its Node32 grid
    s->h_meshes.clear();
    s->stack_entries = 1;
    const TreeOpt topt = tree_opt_settings();
    s->trees_optimized = 0;
    for (HostMesh& m : s->meshes) {
        FlatMesh fm; std::string err;
        // The handed-over tree is restructured before it is laid out (hits do not depend on the tree): SAH re-partition
        // of the host's leaves + insertion rounds; a library-built tree gets the insertion rounds alone.
        HostTree opt_tree; const HostTree* use = &m.tree;
        if (topt.mode != 0 && m.n_tri >= 4) {
            std::vector<float> leaf((size_t)m.n_tri * 6);
            for (int32_t f = 0; f < m.n_tri; ++f)
                for (int k = 0; k < 3; ++k) {
                    const float a = m.verts[3 * (size_t)m.tri[3 * (size_t)f] + k], b = m.verts[3 * (size_t)m.tri[3 * (size_t)f + 1] + k], c = m.verts[3 * (size_t)m.tri[3 * (size_t)f + 2] + k];
                    leaf[6 * (size_t)f + k] = std::min(a, std::min(b, c)); leaf[6 * (size_t)f + 3 + k] = std::max(a, std::max(b, c));
                }
            opt_tree = m.tree;
            const int passes = m.n_tri <= topt.max_leaves_for_passes ? topt.passes : 0;
            if (tree_optimize(opt_tree, leaf.data(), m.n_tri, HPT_HARD_TREE_DEPTH, HPT_SOFT_TREE_DEPTH, passes, m.host_tree)) { use = &opt_tree; s->trees_optimized++; }
        }
        if (!flatten_mesh(*use, m.verts.data(), m.tri.data(), m.normals.data(), m.ntri.data(), m.uvw.data(), m.ttri.data(), m.n_tri, fm, err))
            return fail(HPT_ERR_INVALID, "%s", err.c_str());
        const int64_t node_base = (int64_t)(nodes.size() / 16), tri_base = (int64_t)(tris.size() / HPT_TRI_FLOATS);
        if (node_base + fm.n_interior >= (1ll << 31) || tri_base + m.n_tri >= (1ll << 31)) return fail(HPT_ERR_LIMIT, "scene: more than 2^31 nodes/triangles");
        auto rebase = [&](int32_t ref) -> int32_t { return ref >= 0 ? (int32_t)(ref + node_base) : ~(int32_t)((~ref) + tri_base); };
        for (int32_t i = 0; i < fm.n_interior; ++i) {
            int32_t l, r;
            std::memcpy(&l, &fm.nodes[(size_t)i * 16 + 3], 4); std::memcpy(&r, &fm.nodes[(size_t)i * 16 + 7], 4);
            l = rebase(l); r = rebase(r);
            std::memcpy(&fm.nodes[(size_t)i * 16 + 3], &l, 4); std::memcpy(&fm.nodes[(size_t)i * 16 + 7], &r, 4);
        }
        DMesh dm; std::memset(&dm, 0, sizeof dm);
        dm.root_ref = rebase(fm.root_ref); dm.tri_base = (int32_t)tri_base; dm.ntri = m.n_tri;
        dm.stack_need = fm.max_depth + 2;   // deferred far children (<= interior depth) + the bottom sentinel
        if (dm.stack_need > 40) return fail(HPT_ERR_LIMIT, "bvh: tree depth %d exceeds the traversal stack limit (40 entries of LDS per lane); pass nodes=NULL to let the library build a balanced tree", fm.max_depth);
        for (int k = 0; k < 3; ++k) { dm.bmin[k] = fm.bmin[k]; dm.bmax[k] = fm.bmax[k]; }
        const QuantFrame qf = quant_frame(fm);
        for (int k = 0; k < 3; ++k) { dm.qbase[k] = qf.base[k]; dm.qscale[k] = qf.scale[k]; }
        trees.push_back({node_base, fm.n_interior, qf});
        s->stack_entries = std::max(s->stack_entries, dm.stack_need);
        s->h_meshes.push_back(dm);
        nodes.insert(nodes.end(), fm.nodes.begin(), fm.nodes.end());
        tris.insert(tris.end(), fm.tris.begin(), fm.tris.end());
        trishade.insert(trishade.end(), fm.trishade.begin(), fm.trishade.end());
    }
    // ---- instances in the reference's visiting order (scene.rs:193, :245-248): object first, then children
    std::vector<DInstance> inst; std::vector<DXformInv> xinv; std::vector<DXformFwd> xfwd; std::vector<int32_t> meshinst;
    {
        struct Frame { int32_t node; size_t next_child; };
        int32_t prev_top = -1;
        for (int32_t topid : s->top) {
            std::vector<Frame> stack{{topid, 0}};
            std::vector<int32_t> chain{topid};
            auto emit = [&](int32_t id) -> int {
                const HostNode& n = s->nodes[id];
                if (n.geom == HPT_GEOM_NONE) return HPT_OK;
                DInstance di; std::memset(&di, 0, sizeof di);
                di.geom = n.geom; di.mesh = n.mesh; di.material = n.material; di.node_id = id;
                di.xf_first = (int32_t)xinv.size(); di.xf_count = (int32_t)chain.size();
                di.shared_top = (chain[0] == prev_top) ? 1 : 0;
                prev_top = chain[0];
                int level = 0;
                for (int32_t cid : chain) {
                    const HostNode& cn = s->nodes[cid];
                    float inv[9];
                    if (!mat3_invert(cn.m, inv)) return fail(HPT_ERR_INVALID, "node transform is singular");
                    DXformInv xi; DXformFwd xf;
                    float* ir[3] = {xi.r0, xi.r1, xi.r2}; float* mr[3] = {xf.m0, xf.m1, xf.m2}; float* nr[3] = {xf.n0, xf.n1, xf.n2};
                    for (int r = 0; r < 3; ++r) {
                        for (int c = 0; c < 3; ++c) { ir[r][c] = inv[c * 3 + r]; mr[r][c] = cn.m[c * 3 + r]; nr[r][c] = inv[r * 3 + c]; }
                        ir[r][3] = cn.t[r]; mr[r][3] = cn.t[r]; nr[r][3] = 0.0f;
                    }
                    if (level < 2) di.lv[level] = xi;
                    ++level;
                    xinv.push_back(xi); xfwd.push_back(xf);
                }
                if (di.geom == HPT_GEOM_MESH) meshinst.push_back((int32_t)inst.size());
                inst.push_back(di);
                return HPT_OK;
            };
            CHK(emit(topid));
            while (!stack.empty()) {
                Frame& f = stack.back();
                const HostNode& n = s->nodes[f.node];
                if (f.next_child < n.children.size()) {
                    const int32_t c = n.children[f.next_child++];
                    chain.push_back(c);
                    CHK(emit(c));
                    stack.push_back({c, 0});
                } else { stack.pop_back(); chain.pop_back(); }
            }
        }
    }
    if (inst.size() >= 0x3fffffffu) return fail(HPT_ERR_LIMIT, "too many instances");
    if (tris.size() / HPT_TRI_FLOATS >= 0x40000000ull) return fail(HPT_ERR_LIMIT, "scene: more than 2^30 triangles");
    // ---- TLAS over the instances' world boxes when the linear instance loop of scene.rs:193 would dominate
    int tlas_min = 24;
    if (const char* e = getenv("HPT_TLAS_MIN")) tlas_min = atoi(e);
    int32_t use_tlas = 0, tlas_root = 0;
    int32_t inst_base = 0x40000000, analytic_base = 0x60000000, restore_ref = (int32_t)~0x7ffffffe, stk16_ok = 0;
    s->tlas_nodes = 0;
    QuantFrame tlas_qf; std::memset(&tlas_qf, 0, sizeof tlas_qf);
    // padded world boxes of the instances: leaves of the TLAS
    std::vector<float> wb(inst.size() * 6);
    {
        for (size_t i = 0; i < inst.size(); ++i) {
            const DInstance& di = inst[i];
            double lo[3], hi[3];
            if (di.geom == HPT_GEOM_SPHERE) { for (int k = 0; k < 3; ++k) { lo[k] = -1; hi[k] = 1; } }
            else if (di.geom == HPT_GEOM_PLANE) { lo[0] = lo[1] = -1; hi[0] = hi[1] = 1; lo[2] = hi[2] = 0; }
            else { for (int k = 0; k < 3; ++k) { lo[k] = s->h_meshes[di.mesh].bmin[k]; hi[k] = s->h_meshes[di.mesh].bmax[k]; } }
            double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300}, scale = 0.0;
            for (int c = 0; c < 8; ++c) {
                double p[3] = {(c & 1) ? hi[0] : lo[0], (c & 2) ? hi[1] : lo[1], (c & 4) ? hi[2] : lo[2]};
                for (int k = di.xf_count - 1; k >= 0; --k) {   // from_local through the chain, innermost first (scene.rs:282)
                    const DXformFwd& X = xfwd[di.xf_first + k];
                    const double q[3] = {X.m0[0] * p[0] + X.m0[1] * p[1] + X.m0[2] * p[2] + X.m0[3],
                                         X.m1[0] * p[0] + X.m1[1] * p[1] + X.m1[2] * p[2] + X.m1[3],
                                         X.m2[0] * p[0] + X.m2[1] * p[1] + X.m2[2] * p[2] + X.m2[3]};
                    p[0] = q[0]; p[1] = q[1]; p[2] = q[2];
                }
                for (int k = 0; k < 3; ++k) { mn[k] = std::min(mn[k], p[k]); mx[k] = std::max(mx[k], p[k]); scale = std::max(scale, std::fabs(p[k])); }
            }
            // conservative: the f32 chain of the kernels and this f64 evaluation differ by rounding
            double ext = std::max({mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2]});
            const double pad = 1e-4 * ext + 1e-5 * scale + 1e-30;
            for (int k = 0; k < 3; ++k) { wb[6 * i + k] = (float)(mn[k] - pad); wb[6 * i + 3 + k] = (float)(mx[k] + pad); }
            // (float) rounds to nearest: push the box outward to the next float so that it contains the f64 one
            for (int k = 0; k < 3; ++k) {
                if ((double)wb[6 * i + k] > mn[k] - pad) wb[6 * i + k] = std::nextafterf(wb[6 * i + k], -INFINITY);
                if ((double)wb[6 * i + 3 + k] < mx[k] + pad) wb[6 * i + 3 + k] = std::nextafterf(wb[6 * i + 3 + k], INFINITY);
            }
        }
    }
    // (mesh-queue items of the instance-list path carry the mesh and the position in mesh_instances[] in 16 bits each)
    const bool list_limits = meshinst.size() >= 65536 || s->h_meshes.size() >= 65536;
    if (((int)inst.size() >= tlas_min || list_limits) && inst.size() >= 2) {      // (>= 2: the root of the TLAS is an interior node)
        HostTree tt; FlatMesh ft; std::string err;
        tree_build_sah_boxes(wb.data(), (int32_t)inst.size(), tt);
        if (topt.mode != 0 && (int32_t)inst.size() <= topt.max_leaves_for_passes)
            (void)tree_optimize(tt, wb.data(), (int32_t)inst.size(), 24, 24, topt.passes, false);
        if (!flatten_tree(tt, wb.data(), (int32_t)inst.size(), ft, err)) return fail(HPT_ERR_INVALID, "%s", err.c_str());
        const int64_t node_base = (int64_t)(nodes.size() / 16);
        if (node_base + ft.n_interior >= (1ll << 31)) return fail(HPT_ERR_LIMIT, "scene: more than 2^31 nodes");
        // leaf refs: ~(inst_base + i) for a mesh instance, ~(analytic_base + i) for a sphere / plane (the kernel may defer those);
        // codes packed from 0 when they and the interior refs fit a signed 16-bit stack entry (hpt_device.h DScene.inst_base)
        if (inst.size() >= (1u << 26)) return fail(HPT_ERR_LIMIT, "scene: more than 2^26 instances");
        const uint64_t n_tri_all = tris.size() / HPT_TRI_FLOATS, n_codes = n_tri_all + 2 * (uint64_t)inst.size() + 1;
        // ... for the scenes whose traversal tests everything in place (analytic instances below a quarter, triangle deferral off):
        // that kernel variant then runs with a 16-bit stack at five waves per SIMD; the deferring variants are register-bound at four
        size_t n_analytic = 0;
        for (const DInstance& di : inst) n_analytic += di.geom != HPT_GEOM_MESH;
        const char* e16 = getenv("HPT_TLAS_STK16"); const char* edef = getenv("HPT_TLAS_TRI_DEFER");
        const bool want16 = !(e16 && atoi(e16) == 0) && !(edef && atoi(edef) != 0) && n_analytic * 4 < inst.size();
        if (want16 && n_codes < 32767 && node_base + ft.n_interior < 32767) {
            inst_base = (int32_t)n_tri_all; analytic_base = inst_base + (int32_t)inst.size(); restore_ref = ~(analytic_base + (int32_t)inst.size()); stk16_ok = 1;
        }
        auto rebase = [&](int32_t ref) -> int32_t {
            if (ref >= 0) return (int32_t)(ref + node_base);
            const int32_t i = ~ref;
            return ~(int32_t)((inst[(size_t)i].geom == HPT_GEOM_MESH ? inst_base : analytic_base) + i);
        };
        for (int32_t i = 0; i < ft.n_interior; ++i) {
            int32_t l, r;
            std::memcpy(&l, &ft.nodes[(size_t)i * 16 + 3], 4); std::memcpy(&r, &ft.nodes[(size_t)i * 16 + 7], 4);
            l = rebase(l); r = rebase(r);
            std::memcpy(&ft.nodes[(size_t)i * 16 + 3], &l, 4); std::memcpy(&ft.nodes[(size_t)i * 16 + 7], &r, 4);
        }
        nodes.insert(nodes.end(), ft.nodes.begin(), ft.nodes.end());
        tlas_qf = quant_frame(ft);
        trees.push_back({node_base, ft.n_interior, tlas_qf});
        use_tlas = 1; tlas_root = rebase(ft.root_ref); s->tlas_nodes = ft.n_interior;
        // one stack: TLAS levels + the restore marker + the deepest BLAS + the bottom sentinel
        s->stack_entries = ft.max_depth + 1 + s->stack_entries + 1;
        if (s->stack_entries > 56) return fail(HPT_ERR_LIMIT, "two-level traversal needs %d stack entries (limit 56)", s->stack_entries);
    }
    // ---- materials, images
    std::vector<DMaterial> mats(s->materials.size());
    for (size_t i = 0; i < mats.size(); ++i) {
        const hpt_material& m = s->materials[i];
        conv_texture(m.diffuse, mats[i].diffuse); conv_texture(m.reflection, mats[i].reflection); conv_texture(m.refraction, mats[i].refraction);
        for (int k = 0; k < 3; ++k) mats[i].emission[k] = m.emission[k];
        mats[i].refl_gloss = m.reflection_glossiness; mats[i].refr_gloss = m.refraction_glossiness; mats[i].ior = m.refraction_index;
        mats[i].cos_refl_gloss = cosf(m.reflection_glossiness); mats[i].cos_refr_gloss = cosf(m.refraction_glossiness);
        { volatile float n1 = 1.0f, n2 = m.refraction_index; volatile float q = (n1 - n2) / (n1 + n2); mats[i].r0 = q * q; }   // same value from either side
    }
    std::vector<DImage> imgs(s->images.size()); std::vector<uint8_t> pool;
    for (size_t i = 0; i < imgs.size(); ++i) {
        imgs[i].offset = pool.size(); imgs[i].w = s->images[i].w; imgs[i].h = s->images[i].h;
        pool.insert(pool.end(), s->images[i].px.begin(), s->images[i].px.end());
        while (pool.size() % 16) pool.push_back(0);
    }
    // ---- node format.  Node32 (16-bit child boxes, 2 quads per node) halves the bytes AND the load instructions of every
    // traversal step (two dwordx4 per lane instead of four) at the price of 12 conversions (v_cvt_f32_u32 with a word
    // select).  Round 1 measured +-1 % on the L2-resident teapot, whose traversal kernel was bound by instruction issue
    // then; since round 2 that kernel is bound by its dependent fetches and the vector-memory front end, and Node32 wins on
    // every workload (r02g: teapot +3 %, 3 375 instanced teapots +4.7 %, 10 k spheres +4.3 %, 1 M-triangle soup +31 %; node
    // visits per ray unchanged to the third digit).  So: Node32 whenever every tree's grid resolves its leaves (outward
    // rounding grows a box by <= 4 quanta per axis), else Node64.  HPT_NODE_FORMAT=64 / 32 forces one.  Results do
    // not depend on it: boxes only grow, hits are decided by the exact triangle / sphere / plane tests.
    bool node32 = true;
    for (const TreeSlice& t : trees) if (t.n_interior > 0 && t.qf.quantum * 64.0 > t.qf.mean_leaf_extent) node32 = false;
    if (const char* e = getenv("HPT_NODE_FORMAT")) { if (atoi(e) == 64) node32 = false; else if (atoi(e) == 32) node32 = true; }
    std::vector<uint32_t> nodes32;
    if (node32) {
        nodes32.resize(nodes.size() / 2);
        for (const TreeSlice& t : trees) {
            std::vector<uint32_t> q; std::string err;
            if (!quantize_nodes(nodes.data() + (size_t)t.node_base * 16, t.n_interior, t.qf, q, err)) { node32 = false; break; }
            std::memcpy(nodes32.data() + (size_t)t.node_base * 8, q.data(), q.size() * sizeof(uint32_t));
        }
    }
    if (node32) CHK(upload(s->d_nodes, nodes32, st)); else CHK(upload(s->d_nodes, nodes, st));
    s->node_bytes = node32 ? 32 : 64;
    CHK(upload(s->d_tris, tris, st)); CHK(upload(s->d_trishade, trishade, st));
    CHK(upload(s->d_meshes, s->h_meshes, st)); CHK(upload(s->d_instances, inst, st)); CHK(upload(s->d_meshinst, meshinst, st));
    CHK(upload(s->d_xinv, xinv, st)); CHK(upload(s->d_xfwd, xfwd, st));
    // world-space normal of every plane instance: (0, 0, 1) carried out through the chain with the f32 operations of world_hit (scene.rs:286
    // per level: Minv^T n, normalised) -- the default shade reads it instead of redoing the chain per hit
    std::vector<float> pnormal(inst.size() * 4, 0.0f);
    for (size_t i = 0; i < inst.size(); ++i) {
        if (inst[i].geom != HPT_GEOM_PLANE) continue;
        float n[3] = {0.0f, 0.0f, 1.0f};
        for (int k = inst[i].xf_count - 1; k >= 0; --k) {
            const DXformFwd& X = xfwd[inst[i].xf_first + k];
            const float a = (X.n0[0] * n[0] + X.n0[1] * n[1]) + X.n0[2] * n[2], b = (X.n1[0] * n[0] + X.n1[1] * n[1]) + X.n1[2] * n[2],
                        c = (X.n2[0] * n[0] + X.n2[1] * n[1]) + X.n2[2] * n[2];
            const float inv = 1.0f / std::sqrt((a * a + b * b) + c * c);
            n[0] = a * inv; n[1] = b * inv; n[2] = c * inv;
        }
        for (int k = 0; k < 3; ++k) pnormal[4 * i + k] = n[k];
    }
    CHK(upload(s->d_pnormal, pnormal, st));
    CHK(upload(s->d_materials, mats, st)); CHK(upload(s->d_images, imgs, st)); CHK(upload(s->d_pool, pool, st));
    HIPCHK(hipStreamSynchronize(st));   // the staging vectors die at scope end
    DScene& d = s->ds;
    d.nodes = s->d_nodes.as<float4>(); d.tris = s->d_tris.as<float4>(); d.trishade = s->d_trishade.as<float4>();
    d.meshes = s->d_meshes.as<DMesh>(); d.instances = s->d_instances.as<DInstance>(); d.mesh_instances = s->d_meshinst.as<int32_t>();
    d.xinv = s->d_xinv.as<DXformInv>(); d.xfwd = s->d_xfwd.as<DXformFwd>(); d.plane_normal = s->d_pnormal.as<float4>();
    d.materials = s->d_materials.as<DMaterial>(); d.images = s->d_images.as<DImage>(); d.image_pool = s->d_pool.as<uint8_t>();
    d.n_instances = (int32_t)inst.size(); d.n_meshes = (int32_t)s->h_meshes.size();
    s->n_xf = (int)xinv.size(); s->n_mat = (int)mats.size();
    s->n_tris_total = (int64_t)(tris.size() / HPT_TRI_FLOATS);
    s->device_bytes = (node32 ? nodes32.size() : nodes.size()) * 4 + tris.size() * 4 + trishade.size() * 4 + inst.size() * sizeof(DInstance) + xinv.size() * sizeof(DXformInv) +
                      xfwd.size() * sizeof(DXformFwd) + mats.size() * sizeof(DMaterial) + pool.size();
    d.use_tlas = use_tlas; d.tlas_root = tlas_root;
    d.node32 = node32 ? 1 : 0; d.stk16_ok = stk16_ok;
    d.inst_base = inst_base; d.analytic_base = analytic_base; d.restore_ref = restore_ref;
    { const char* e = getenv("HPT_TLAS_TRI_DEFER"); d.tri_defer = (e && atoi(e) != 0) ? 1 : 0; }
    for (int k = 0; k < 3; ++k) { d.tlas_qbase[k] = tlas_qf.base[k]; d.tlas_qscale[k] = tlas_qf.scale[k]; }
    d.n_mesh_instances = (int32_t)meshinst.size(); d.n_nodes = (int32_t)(nodes.size() / 16);
    conv_texture(s->background, d.background); conv_texture(s->environment, d.environment);
    camera_frame(s->camera, d.cam);
    // launch configuration: depends on the committed scene only (occupancy queries, LDS layout) -- not per launch
    hptk_plan(s->stack_entries, d, s->n_xf, s->n_mat, &s->plan);
    // k_shade: two rounds of exactly the resident workgroups (5/CU -> 10/CU measured best: 8/CU +3 %, 4/CU +8 %)
    // instance-list path: k_shade of depth d also runs the analytic stage of depth d + 1 (k_extend_a is then launched for depth 0 only);
    // HPT_FUSE_ANALYTIC=0 at commit keeps the two kernels apart (A/B)
    { const char* e = getenv("HPT_FUSE_ANALYTIC"); s->fuse_analytic = !use_tlas && !(e && atoi(e) == 0); }
    const int shade_per_cu = s->fuse_analytic ? s->plan.shade_fused_per_cu : s->plan.shade_per_cu;
    s->shade_grid = shade_per_cu > 0 ? ctx->num_cus * shade_per_cu * 2 : ctx->max_blocks;
    if (const char* e = getenv("HPT_SHADE_BLOCKS_PER_CU")) { int v = atoi(e); if (v >= 1 && v <= 32) s->shade_grid = ctx->num_cus * v; }
    s->analytic_grid = s->plan.analytic_per_cu > 0 ? ctx->num_cus * s->plan.analytic_per_cu * 2 : ctx->max_blocks;      // (4 / 8 / 16 per CU with two passes in flight: 13.47 / 13.62 / 14.05 Grays/s)
    if (const char* e = getenv("HPT_ANALYTIC_BLOCKS_PER_CU")) { int v = atoi(e); if (v >= 1 && v <= 64) s->analytic_grid = ctx->num_cus * v; }
    static std::atomic<uint64_t> commit_stamp{0};
    s->committed = true; s->generation = ++commit_stamp;
    return HPT_OK;
}

// ------------------------------------------------------------------------------------------------ trace entries
static int need_committed(hpt_scene* s) {
    if (!s) return fail(HPT_ERR_INVALID, "null scene");
    if (!s->committed) return fail(HPT_ERR_STATE, "scene not committed: call hpt_scene_commit first");
    return HPT_OK;
}

extern "C" int hpt_trace_rays_device(hpt_scene* s, int32_t mesh_id, int64_t n, const void* org4, const void* dir4,
                                     void* hit4, void* side_out, int count_visits, hpt_trace_stats* stats) {
    CHK(need_committed(s));
    if (mesh_id < 0 || mesh_id >= (int32_t)s->h_meshes.size()) return fail(HPT_ERR_INVALID, "hpt_trace_rays: mesh id %d does not exist", mesh_id);
    if (n < 0) return fail(HPT_ERR_INVALID, "hpt_trace_rays: negative ray count");
    if (n > 0 && (!org4 || !dir4 || !hit4 || !side_out)) return fail(HPT_ERR_INVALID, "hpt_trace_rays: null pointer");
    hpt_context* ctx = s->ctx;
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    if (stats) std::memset(stats, 0, sizeof *stats);
    if (n == 0) return HPT_OK;
    CHK(s->w_stat.ensure(8 * sizeof(unsigned long long)));
    unsigned long long* d_stat = s->w_stat.as<unsigned long long>();
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (stats) {
        HIPCHK(hipMemsetAsync(d_stat, 0, 8 * sizeof(unsigned long long), st));
        // two events of the context's pool (created once, destroyed with the context: nothing to leak on an error return)
        while (ctx->events.size() < 2) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); ctx->events.push_back(e); }
        e0 = ctx->events[0]; e1 = ctx->events[1];
        HIPCHK(hipEventRecord(e0, st));
    }
    // mesh queue of the root-box pass: items for one 2^30-ray launch at a time, one counter (its own 128-B line) per launch
    const size_t launches = (size_t)((n + (1ll << 30) - 1) >> 30);
    CHK(s->t_mq.ensure(hptk_mq_slots((uint64_t)std::min<int64_t>(n, 1ll << 30)) * 8)); CHK(s->t_mqcount.ensure(launches * HPT_COUNTS_PER_DEPTH * 4));
    HIPCHK(hipMemsetAsync(s->t_mqcount.p, 0, launches * HPT_COUNTS_PER_DEPTH * 4, st));
    hptk_trace_rays(st, ctx->extend_blocks, s->ds, s->h_meshes[mesh_id], n, (const float4*)org4, (const float4*)dir4, (float4*)hit4, (uint8_t*)side_out,
                    count_visits != 0, d_stat, s->t_mq.as<uint2>(), s->t_mqcount.as<uint32_t>());
    HIPCHK(hipGetLastError());
    if (stats) {
        HIPCHK(hipEventRecord(e1, st));
        HIPCHK(hipStreamSynchronize(st));
        float ms = 0.0f;
        HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long h[8];
        HIPCHK(hipMemcpy(h, d_stat, sizeof h, hipMemcpyDeviceToHost));
        stats->rays = (uint64_t)n; stats->nodes_visited = h[1]; stats->tris_tested = h[2];
        stats->seconds_kernel = ms * 1e-3; stats->seconds_total = ms * 1e-3;
    }
    return HPT_OK;
}

static void pack4(const float* src3, int64_t n, std::vector<float>& dst) {
    dst.resize((size_t)n * 4);
    for (int64_t i = 0; i < n; ++i) { dst[4 * i] = src3[3 * i]; dst[4 * i + 1] = src3[3 * i + 1]; dst[4 * i + 2] = src3[3 * i + 2]; dst[4 * i + 3] = 0.0f; }
}

extern "C" int hpt_trace_rays(hpt_scene* s, int32_t mesh_id, int64_t n, const float* org3, const float* dir3,
                              int32_t* prim, float* t, float* u, float* v, uint8_t* side, hpt_trace_stats* stats) {
    CHK(need_committed(s));
    if (n < 0) return fail(HPT_ERR_INVALID, "hpt_trace_rays: negative ray count");
    if (n > 0 && (!org3 || !dir3 || !prim || !t || !u || !v || !side)) return fail(HPT_ERR_INVALID, "hpt_trace_rays: null pointer");
    if (mesh_id < 0 || mesh_id >= (int32_t)s->h_meshes.size()) return fail(HPT_ERR_INVALID, "hpt_trace_rays: mesh id %d does not exist", mesh_id);
    const double t0 = now_s();
    hpt_trace_stats total; std::memset(&total, 0, sizeof total);
    hpt_context* ctx = s->ctx;
    HIPCHK(hipSetDevice(ctx->device));
    const int64_t CH = 1 << 22;
    std::vector<float> o4, d4, h4; std::vector<uint8_t> sd;
    for (int64_t off = 0; off < n; off += CH) {
        const int64_t m = std::min(CH, n - off);
        pack4(org3 + 3 * off, m, o4); pack4(dir3 + 3 * off, m, d4);
        CHK(s->t_org.ensure((size_t)m * 16)); CHK(s->t_dir.ensure((size_t)m * 16)); CHK(s->t_hit.ensure((size_t)m * 16)); CHK(s->t_side.ensure((size_t)m));
        HIPCHK(hipMemcpyAsync(s->t_org.p, o4.data(), (size_t)m * 16, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(s->t_dir.p, d4.data(), (size_t)m * 16, hipMemcpyHostToDevice, ctx->stream));
        hpt_trace_stats cs;
        CHK(hpt_trace_rays_device(s, mesh_id, m, s->t_org.p, s->t_dir.p, s->t_hit.p, s->t_side.p, stats ? 1 : 0, &cs));
        h4.resize((size_t)m * 4); sd.resize((size_t)m);
        HIPCHK(hipMemcpy(h4.data(), s->t_hit.p, (size_t)m * 16, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(sd.data(), s->t_side.p, (size_t)m, hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < m; ++i) {
            int32_t p; std::memcpy(&p, &h4[4 * i + 3], 4);
            prim[off + i] = p; t[off + i] = h4[4 * i]; u[off + i] = h4[4 * i + 1]; v[off + i] = h4[4 * i + 2]; side[off + i] = sd[i];
        }
        total.rays += cs.rays; total.nodes_visited += cs.nodes_visited; total.tris_tested += cs.tris_tested; total.seconds_kernel += cs.seconds_kernel;
    }
    total.seconds_total = now_s() - t0;
    if (stats) *stats = total;
    return HPT_OK;
}

extern "C" int hpt_scene_intersect(hpt_scene* s, int64_t n, const float* org3, const float* dir3,
                                   int32_t* node, int32_t* prim, float* t, uint8_t* side, float* pos3, float* normal3, float* uvw3) {
    CHK(need_committed(s));
    if (n < 0) return fail(HPT_ERR_INVALID, "hpt_scene_intersect: negative ray count");
    if (n == 0) return HPT_OK;
    if (!org3 || !dir3) return fail(HPT_ERR_INVALID, "hpt_scene_intersect: null pointer");
    if (n > (1ll << 26)) return fail(HPT_ERR_LIMIT, "hpt_scene_intersect: at most 2^26 rays per call");
    hpt_context* ctx = s->ctx;
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    std::vector<float> o4, d4;
    pack4(org3, n, o4); pack4(dir3, n, d4);
    const size_t N = (size_t)n;
    CHK(s->t_org.ensure(N * 16)); CHK(s->t_dir.ensure(N * 16));
    const size_t sizes[9] = {N * 4, N * 4, N * 4, N, N * 16, N * 16, N * 16, N * 16, N * 4};
    for (int k = 0; k < 9; ++k) CHK(s->t_misc[k].ensure(sizes[k]));
    CHK(s->t_mq.ensure(hptk_mq_slots(N) * 8)); CHK(s->t_mqcount.ensure((size_t)HPT_COUNTS_PER_DEPTH * 4));
    HIPCHK(hipMemsetAsync(s->t_mqcount.p, 0, (size_t)HPT_COUNTS_PER_DEPTH * 4, st));
    HIPCHK(hipMemcpyAsync(s->t_org.p, o4.data(), N * 16, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(s->t_dir.p, d4.data(), N * 16, hipMemcpyHostToDevice, st));
    hptk_scene_intersect(st, ctx->extend_blocks, s->stack_entries, s->ds, n, s->t_org.as<float4>(), s->t_dir.as<float4>(), s->t_misc[7].as<float4>(),
                         s->t_misc[8].as<uint32_t>(), s->t_misc[0].as<int>(),
                         s->t_misc[1].as<int>(), s->t_misc[2].as<float>(), s->t_misc[3].as<uint8_t>(), s->t_misc[4].as<float4>(),
                         s->t_misc[5].as<float4>(), s->t_misc[6].as<float4>(), s->t_mq.as<uint2>(), s->t_mqcount.as<uint32_t>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    if (node) HIPCHK(hipMemcpy(node, s->t_misc[0].p, N * 4, hipMemcpyDeviceToHost));
    if (prim) HIPCHK(hipMemcpy(prim, s->t_misc[1].p, N * 4, hipMemcpyDeviceToHost));
    if (t) HIPCHK(hipMemcpy(t, s->t_misc[2].p, N * 4, hipMemcpyDeviceToHost));
    if (side) HIPCHK(hipMemcpy(side, s->t_misc[3].p, N, hipMemcpyDeviceToHost));
    std::vector<float> tmp(N * 4);
    float* outs[3] = {pos3, normal3, uvw3};
    for (int k = 0; k < 3; ++k) {
        if (!outs[k]) continue;
        HIPCHK(hipMemcpy(tmp.data(), s->t_misc[4 + k].p, N * 16, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < N; ++i) { outs[k][3 * i] = tmp[4 * i]; outs[k][3 * i + 1] = tmp[4 * i + 1]; outs[k][3 * i + 2] = tmp[4 * i + 2]; }
    }
    return HPT_OK;
}

// ------------------------------------------------------------------------------------------------ render driver
struct EventLog {   // HPT_RENDER_TIME_KERNELS: (kind, start, stop) triples from the context's event pool
    hpt_context* ctx; bool on; size_t used = 0;
    struct Rec { int kind; size_t a, b; };
    std::vector<Rec> recs;
    int begin(int kind, hipStream_t st) {
        if (!on) return HPT_OK;
        while (ctx->events.size() < used + 2) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); ctx->events.push_back(e); }
        HIPCHK(hipEventRecord(ctx->events[used], st));
        recs.push_back({kind, used, used + 1});
        return HPT_OK;
    }
    int end(hipStream_t st) {
        if (!on) return HPT_OK;
        HIPCHK(hipEventRecord(ctx->events[used + 1], st));
        used += 2;
        return HPT_OK;
    }
    int collect(double out[5], uint32_t* n_extend) {
        for (int k = 0; k < 5; ++k) out[k] = 0.0;
        for (const Rec& r : recs) {
            float ms = 0.0f;
            HIPCHK(hipEventElapsedTime(&ms, ctx->events[r.a], ctx->events[r.b]));
            out[r.kind] += ms * 1e-3;
            if (r.kind == 1 && n_extend) (*n_extend)++;
        }
        return HPT_OK;
    }
};

static int default_spp_per_pass(uint64_t npix, int want) {
    if (want > 0) return want > 64 ? 64 : want;
    const uint64_t target_paths = 64ull << 20;   // ~64M paths in flight per pass (8.9 GB of queues): 8 M -> 32 M +5 %, 64 M +1 %, 128 M -0.5 %
    uint64_t s = target_paths / std::max<uint64_t>(npix, 1);
    if (s < 1) s = 1;
    if (s > 64) s = 64;   // the sample slot of a path travels in 6 bits next to its pixel index
    return (int)s;
}

// one wavefront pass: raygen -> [extend -> shade]* -> resolve, all asynchronous on the context stream
// Persistent grid of k_extend: exactly the workgroups the device keeps resident (registers and LDS of the variant that
// runs: the occupancy query) -- work is strided statically over the grid, so a workgroup that had to wait for a free
// slot would start its share when the others are done.
// With two passes in flight the grid leaves one workgroup slot per CU free: k_extend's LDS would otherwise fill the CU,
// and the other pass's k_shade (4 KB of LDS tables, 72 registers) could only start where a k_extend workgroup has ended.
// Beside four k_extend workgroups two k_shade workgroups per CU fit, and their arithmetic fills issue slots the
// traversal leaves empty while it waits for nodes (measured: 11.7 -> 12.0 Grays/s; on one stream 4 per CU costs 4 %).
static int extend_grid(const hpt_scene* s, int inflight) {
    const hpt_context* ctx = s->ctx;
    int per_cu = ctx->extend_per_cu;
    if (per_cu <= 0) {
        per_cu = s->plan.extend_per_cu;
        if (per_cu <= 0)      // query failed: register budget of the kernels (hpt_kernels.hip) and LDS fit
            per_cu = std::min(4, (int)((160u * 1024u) / std::max<size_t>(s->plan.extend_lds, 1)));
        // (two-level kernel: one slot -- r03c, HPT_EXTEND_BLOCKS_PER_CU sweep with two passes in flight: 3 375 instanced teapots
        //  2 / 3 / 4 / 5 per CU of 5 resident = 2.86 / 3.17 / 3.21 / 3.11 Grays/s, 10 k spheres 2 / 3 / 4 of 4 = 4.19 / 4.27 / 4.15)
        // (instance-list path: every resident slot since k_shade carries the analytic stage -- r03e, 1080p teapot, two passes in flight:
        //  4 / 6 / 7 of 7 per CU = 17.97 / 18.45 / 18.49 Grays/s; r03b, with the analytic stage a kernel of its own, had 4 ahead; the soup does not care)
        // (instance-list path with a big tree -- the 1 M-triangle soup, bandwidth-bound, fed through the work counter: HALF the resident slots, so that
        //  the traversals of both passes in flight are resident side by side -- r03f, 4 resident: 1 / 2 / 3 / 4 per CU = 1.63 / 2.14 / 2.12 / 2.04 Grays/s,
        //  while on one stream the kernel itself saturates at 4: 19.8 / 11.0 / 8.7 / 8.3 ms per launch)
        const int spare = ctx->extend_spare >= 0 ? ctx->extend_spare : (s->ds.use_tlas ? 1 : (s->plan.extend_counter ? per_cu / 2 : 0));
        if (inflight > 1 && per_cu > spare) per_cu -= spare;
    }
    return ctx->num_cus * std::max(1, per_cu);
}

// `st`: the stream of this pass; `after` (may be null): the resolve of the previous pass, on the other stream -- the
// resolves of consecutive passes read and write the same accumulator, so they are chained (the sums themselves do not
// depend on the order: 64-bit fixed point); `done` (may be null) is recorded behind this pass's resolve.
static int run_pass(hpt_scene* s, DPass& ps, long long* accum, float* minmax, bool counters, EventLog& ev, int extend_blocks,
                    hipStream_t st, hipEvent_t after, hipEvent_t done) {
    hpt_context* ctx = s->ctx;
    HIPCHK(hipMemsetAsync(ps.counts, 0, (size_t)HPT_COUNTS_TOTAL * sizeof(uint32_t), st));
    HIPCHK(hipMemsetAsync(ps.radiance, 0, (size_t)ps.n_slots * ps.S * sizeof(float4), st));
    const bool fused = s->fuse_analytic && ps.hit_par != 0;      // raygen / k_shade run the analytic stage of the ray they make
    CHK(ev.begin(0, st)); hptk_raygen(st, ctx->max_blocks, s->ds, ps, fused, counters); CHK(ev.end(st));
    const uint64_t n_paths = (uint64_t)ps.n_slots * ps.S;
    const int blocks = (int)std::min<uint64_t>((n_paths + HPT_BLOCK - 1) / HPT_BLOCK, (uint64_t)extend_blocks);
    const int sblocks = (int)std::min<uint64_t>((n_paths + HPT_BLOCK - 1) / HPT_BLOCK, (uint64_t)s->shade_grid);
    // k_extend_a: at least 32 chunks per workgroup before the grid grows beyond 4 per CU -- a workgroup that finds nothing
    // to do still costs its launch and its look at the 64 segment counters, and at the deep depths of a small pass most
    // of a big grid does (8 M-path passes, 16 spp: 9.1 ms with 16 per CU, 6.9 ms with 4)
    const uint64_t chunks = (n_paths + HPT_BLOCK - 1) / HPT_BLOCK;
    const int ablocks = (int)std::min<uint64_t>(std::min<uint64_t>(chunks, std::max<uint64_t>(chunks / 32, (uint64_t)ctx->num_cus * 4)), (uint64_t)s->analytic_grid);
    for (int depth = 0; depth <= ps.max_depth; ++depth) {
        if (!s->ds.use_tlas) {      // instance-list path: analytic stage over every ray, then the traversal over the rays that entered a mesh box
            if (!fused) { CHK(ev.begin(4, st)); hptk_extend_analytic(st, ablocks, s->ds, ps, depth, counters); CHK(ev.end(st)); }      // (fused: k_raygen / k_shade of depth - 1 did it)
            if (s->ds.n_mesh_instances > 0) {
                // (timed with the analytic stage: seconds_analytic = everything between the rays and the traversal)
                if (ps.mq_bin && depth >= 1) { CHK(ev.begin(4, st)); hptk_mq_bin(st, ablocks, s->ds, ps, depth); CHK(ev.end(st)); }
                CHK(ev.begin(1, st)); hptk_extend(st, blocks, s->plan, s->ds, ps, depth, counters); CHK(ev.end(st));
            }
        } else { CHK(ev.begin(1, st)); hptk_extend(st, blocks, s->plan, s->ds, ps, depth, counters); CHK(ev.end(st)); }
        CHK(ev.begin(2, st)); hptk_shade(st, sblocks, s->plan, s->ds, ps, depth, s->n_xf, s->n_mat, fused, counters, (ps.flags & HPT_RENDER_EXACT_SHADE) != 0); CHK(ev.end(st));
    }
    if (after) HIPCHK(hipStreamWaitEvent(st, after, 0));
    CHK(ev.begin(3, st)); hptk_resolve(st, ctx->max_blocks, ps, accum, minmax); CHK(ev.end(st));
    hptk_finish_pass(st, ps);
    HIPCHK(hipGetLastError());
    if (done) HIPCHK(hipEventRecord(done, st));
    return HPT_OK;
}

static uint32_t shard_cap_for(uint64_t n_paths) {
    const uint64_t chunks = (n_paths + HPT_BLOCK - 1) / HPT_BLOCK;
    return (uint32_t)(((chunks + HPT_SHARDS - 1) / HPT_SHARDS) * HPT_BLOCK);
}
static int ensure_pass_workspace(hpt_scene* s, uint64_t n_paths, DPass& ps, int set) {
    if (n_paths >= (1ull << 31) - (1ull << 20)) return fail(HPT_ERR_LIMIT, "render: more than 2^31 paths per pass; lower spp_per_pass");
    const uint64_t max_paths = (uint64_t)shard_cap_for(n_paths) * HPT_SHARDS;   // slots of a sharded queue
    DevBuf (&q)[2][3] = set ? s->x[set - 1].q : s->w_q;
    DevBuf &hit = set ? s->x[set - 1].hit : s->w_hit, &hit_inst = set ? s->x[set - 1].hit_inst : s->w_hit_inst, &mq = set ? s->x[set - 1].mq : s->w_mq,
           &radiance = set ? s->x[set - 1].radiance : s->w_radiance, &counts = set ? s->x[set - 1].counts : s->w_counts;
    for (int a = 0; a < 2; ++a) for (int k = 0; k < 3; ++k) CHK(q[a][k].ensure(max_paths * 16));
    const uint64_t hit_sets = s->fuse_analytic ? 2 : 1;      // fused shade + analytic stage: hit records of even and odd depths apart
    CHK(hit.ensure(hit_sets * max_paths * 16)); CHK(hit_inst.ensure(hit_sets * max_paths * 4)); CHK(radiance.ensure(max_paths * 16));
    ps.hit_par = s->fuse_analytic ? (uint32_t)max_paths : 0u;
    if (!s->ds.use_tlas && s->ds.n_mesh_instances > 0) CHK(mq.ensure(max_paths * 48));      // mesh queue of the instance-list path: items, origins, directions
    DevBuf& mq_bin = set ? s->x[set - 1].mq_bin : s->w_mq_bin;
    ps.mq_bin = nullptr;
    if (hptk_wants_binning(s->ds)) { CHK(mq_bin.ensure(max_paths * 48)); ps.mq_bin = mq_bin.as<uint4>(); }      // big trees: the same queue in cell order
    CHK(counts.ensure((size_t)HPT_COUNTS_TOTAL * sizeof(uint32_t)));
    CHK(s->w_stat.ensure(8 * sizeof(unsigned long long)));
    for (int a = 0; a < 2; ++a) { ps.q[a].ray_o = q[a][0].as<float4>(); ps.q[a].ray_d = q[a][1].as<float4>(); ps.q[a].thr = q[a][2].as<float4>(); }
    ps.hit = hit.as<float4>(); ps.hit_inst = hit_inst.as<uint32_t>(); ps.mq = mq.as<uint4>(); ps.radiance = radiance.as<float4>();
    ps.counts = counts.as<uint32_t>(); ps.stat = s->w_stat.as<unsigned long long>();
    return HPT_OK;
}

static int check_params(hpt_scene* s, const hpt_render_params* p) {
    CHK(need_committed(s));
    if (!p) return fail(HPT_ERR_INVALID, "null render params");
    if (!s->have_camera) return fail(HPT_ERR_STATE, "no camera set");
    if (p->max_depth < 0 || p->max_depth > HPT_MAX_DEPTH_LIMIT) return fail(HPT_ERR_LIMIT, "max_depth must be in 0..%d", HPT_MAX_DEPTH_LIMIT);
    if (p->adaptive) {
        if (p->initial_samples <= 0 || p->max_samples < p->initial_samples) return fail(HPT_ERR_INVALID, "adaptive: need 0 < initial_samples <= max_samples");
        if (p->max_samples > (1 << 20)) return fail(HPT_ERR_LIMIT, "adaptive: max_samples above 2^20");
    } else {
        if (p->spp <= 0 || p->first_sample < 0) return fail(HPT_ERR_INVALID, "need spp > 0 and first_sample >= 0");
        if ((int64_t)p->spp + p->first_sample > (1 << 24)) return fail(HPT_ERR_LIMIT, "sample index above 2^24");
    }
    if (p->spp_per_pass < 0) return fail(HPT_ERR_INVALID, "spp_per_pass must be >= 0");
    return HPT_OK;
}

// Render of samples first_sample+1 ..= first_sample+spp of the n_active pixels in pixel_list (device; NULL = every
// pixel, n_active = W*H) into accum (device, W*H hpt_accum records, ADDED to) and, when minmax != NULL, folded into the
// per-channel running min/max of main.rs:90-100.
static int render_range(hpt_scene* s, const hpt_render_params* p, const uint32_t* pixel_list, uint32_t n_active, long long* accum, float* minmax,
                        float* per_sample_host, hpt_render_stats* stats, EventLog& ev) {
    hpt_context* ctx = s->ctx;
    hipStream_t st = ctx->stream;
    const uint64_t npix = n_active;
    if (npix == 0) return HPT_OK;
    int S = std::min(default_spp_per_pass(npix, p->spp_per_pass), p->spp);
    // a render that would be ONE big pass runs as two halves, so that two passes can be in flight (results do not depend on the split)
    if (p->spp_per_pass == 0 && S == p->spp && S >= 2 && ctx->overlap > 1 && !per_sample_host && !ev.on && npix * (uint64_t)S >= (16ull << 20))
        S = (S + 1) / 2;
    // ... and a render of few big passes as more, smaller ones while those stay above 32 M paths: the two streams only overlap
    // between the first pass's ramp-up and the last one's drain (1080p, 128 spp -- an eighth of the headline frame, the share
    // of one GPU of eight: 4 passes of 32 spp 160.0 ms, 8 of 16 spp 157.4 ms)
    if (p->spp_per_pass == 0 && ctx->overlap > 1 && !per_sample_host && !ev.on)
        while ((p->spp + S - 1) / S < 8 && S >= 2 && npix * (uint64_t)(S / 2) >= (32ull << 20)) S /= 2;
    // Two passes in flight when there are at least two: pass k runs on stream k & 1 with pass buffers of its own.  The
    // per-kernel event timing of HPT_RENDER_TIME_KERNELS needs launches that have the GPU to themselves: one stream then.
    const int n_passes = (p->spp + S - 1) / S;
    int inflight = (per_sample_host || ev.on) ? 1 : std::min(ctx->overlap, n_passes);
    DPass psv[HPT_MAX_INFLIGHT]; std::memset(psv, 0, sizeof psv);
    CHK(ensure_pass_workspace(s, npix * (uint64_t)S, psv[0], 0));
    for (int k = 1; k < inflight; ++k)
        if (ensure_pass_workspace(s, npix * (uint64_t)S, psv[k], k) != HPT_OK) { inflight = k; (void)hipGetLastError(); break; }   // no memory for another set of pass buffers: fewer in flight
    CHK(s->w_halton.ensure((size_t)p->spp * sizeof(float2)));
    hptk_halton(st, s->w_halton.as<float2>(), (uint32_t)p->first_sample, (uint32_t)p->spp);      // scene.rs:431 on the device: no copy, no synchronisation
    std::vector<float> rad;
    uint32_t passes = 0;
    if (inflight > 1) {
        HIPCHK(hipEventRecord(ctx->ev_fork, st));
        for (int k = 1; k < inflight; ++k) HIPCHK(hipStreamWaitEvent(ctx->aux_stream[k - 1], ctx->ev_fork, 0));
    }
    for (int done = 0; done < p->spp; done += S) {
        const int cur = std::min(S, p->spp - done);
        const int set = (int)(passes % (uint32_t)inflight);
        DPass& ps = psv[set];
        ps.halton = s->w_halton.as<float2>() + done;
        ps.pixel_list = pixel_list; ps.n_slots = (uint32_t)npix; ps.S = (uint32_t)cur;
        ps.first_sample = (uint32_t)(p->first_sample + done); ps.seed = p->seed; ps.max_depth = p->max_depth; ps.flags = p->flags;
        ps.shard_cap = shard_cap_for(npix * (uint64_t)cur);
        const int rc = run_pass(s, ps, accum, minmax, (p->flags & HPT_RENDER_COUNTERS) != 0, ev, extend_grid(s, inflight), set ? ctx->aux_stream[set - 1] : st,
                                inflight > 1 && passes > 0 ? ctx->ev_done[(set + inflight - 1) % inflight] : nullptr, inflight > 1 ? ctx->ev_done[set] : nullptr);
        if (rc != HPT_OK) {
            // the aux streams were forked off the context stream and hold passes of this call: drain them before the error
            // leaves, so that the caller's stream is again the only one with work of this context (g_err keeps rc's text)
            const std::string keep = g_err;
            for (int k = 1; k < inflight; ++k) (void)hipStreamSynchronize(ctx->aux_stream[k - 1]);
            (void)hipStreamSynchronize(st); (void)hipGetLastError();
            g_err = keep;
            return rc;
        }
        ++passes;
        if (per_sample_host) {
            rad.resize((size_t)npix * cur * 4);
            HIPCHK(hipStreamSynchronize(st));
            HIPCHK(hipMemcpy(rad.data(), ps.radiance, rad.size() * sizeof(float), hipMemcpyDeviceToHost));
            for (uint64_t px = 0; px < npix; ++px)
                for (int k = 0; k < cur; ++k)
                    for (int c = 0; c < 3; ++c) per_sample_host[(px * (uint64_t)p->spp + (uint64_t)(done + k)) * 3 + c] = rad[((uint64_t)k * npix + px) * 4 + c];      // radiance is sample-major
        }
    }
    // the last pass's resolve is behind every other one (chain); the context stream continues behind it
    if (inflight > 1 && (passes - 1) % (uint32_t)inflight != 0u) HIPCHK(hipStreamWaitEvent(st, ctx->ev_done[(passes - 1) % (uint32_t)inflight], 0));
    if (stats) { stats->passes += passes; stats->samples += npix * (uint64_t)p->spp; }
    return HPT_OK;
}

static int render_fixed(hpt_scene* s, const hpt_render_params* p, long long* accum, float* per_sample_host, hpt_render_stats* stats, EventLog& ev) {
    return render_range(s, p, nullptr, (uint32_t)((uint64_t)s->ds.cam.w * s->ds.cam.h), accum, nullptr, per_sample_host, stats, ev);
}

static int finish_stats(hpt_scene* s, hpt_render_stats* stats, EventLog& ev, double t0, const void* d_stat = nullptr) {
    if (!stats) return HPT_OK;
    HIPCHK(hipStreamSynchronize(s->ctx->stream));
    unsigned long long h[8];
    HIPCHK(hipMemcpy(h, d_stat ? d_stat : s->w_stat.p, sizeof h, hipMemcpyDeviceToHost));
    stats->rays_primary = h[4]; stats->rays_bounce = h[0] - h[4];
    stats->nodes_visited = h[1]; stats->tris_tested = h[2]; stats->inst_tests = h[3];
    stats->rays_mesh = h[5];
    double k[5];
    CHK(ev.collect(k, &stats->launches_extend));
    stats->seconds_raygen = k[0]; stats->seconds_extend = k[1]; stats->seconds_shade = k[2]; stats->seconds_resolve = k[3]; stats->seconds_analytic = k[4];
    stats->seconds_total = now_s() - t0;
    return HPT_OK;
}

extern "C" int hpt_render_device(hpt_scene* s, const hpt_render_params* p, float* rgb_sum_device, hpt_render_stats* stats) {
    CHK(check_params(s, p));
    if (p->adaptive) return fail(HPT_ERR_INVALID, "hpt_render_device: fixed-spp only (adaptive sampling couples a pixel's samples; use hpt_render)");
    if (!rgb_sum_device) return fail(HPT_ERR_INVALID, "hpt_render_device: null output");
    hpt_context* ctx = s->ctx;
    HIPCHK(hipSetDevice(ctx->device));
    const double t0 = now_s();
    if (stats) std::memset(stats, 0, sizeof *stats);
    const uint64_t npix = (uint64_t)s->ds.cam.w * s->ds.cam.h;
    CHK(s->w_stat.ensure(8 * sizeof(unsigned long long))); CHK(s->w_accum.ensure(npix * sizeof(hpt_accum)));
    HIPCHK(hipMemsetAsync(s->w_stat.p, 0, 8 * sizeof(unsigned long long), ctx->stream));
    HIPCHK(hipMemsetAsync(s->w_accum.p, 0, npix * sizeof(hpt_accum), ctx->stream));
    EventLog ev{ctx, stats && (p->flags & HPT_RENDER_TIME_KERNELS)};
    CHK(render_fixed(s, p, s->w_accum.as<long long>(), nullptr, stats, ev));
    hptk_accum_resolve(ctx->stream, ctx->max_blocks, s->w_accum.as<long long>(), nullptr, rgb_sum_device, (uint32_t)npix, 0.0f);      // the sums as f32
    HIPCHK(hipGetLastError());
    return finish_stats(s, stats, ev, t0);
}

// hpt_render_device for callers that cannot allocate (tight serving loops, harnesses that forbid hipMalloc): the pass
// buffers -- wavefront queues, hit and radiance records, append counters, statistics, Halton table, the call's radiance
// accumulator -- are carved from this device's copy of the static `scratch` array of hpt_kernels.hip (ctx->scratch,
// resolved once per device by hipGetSymbolAddress in hpt_context_create) instead of the scene's grow-only hipMalloc
// workspace.  One pass at a time on the context stream; calls no hipMalloc / hipFree / event or stream creation, no host
// allocation, no host-to-device copy, and -- with stats == NULL -- no synchronisation: it only enqueues kernels and
// memsets (the Halton table is filled by a kernel).  Same kernels and the same fixed-point sums as hpt_render_device ->
// bit-identical output.  Every call on a GPU uses that GPU's one array: overlapping calls race.
extern "C" int hpt_render_device_noalloc(hpt_scene* s, const hpt_render_params* p, float* rgb_sum_device, hpt_render_stats* stats) {
    CHK(check_params(s, p));
    if (p->adaptive) return fail(HPT_ERR_INVALID, "hpt_render_device_noalloc: fixed-spp only");
    if (!rgb_sum_device) return fail(HPT_ERR_INVALID, "hpt_render_device_noalloc: null output");
    if (p->flags & HPT_RENDER_TIME_KERNELS) return fail(HPT_ERR_INVALID, "hpt_render_device_noalloc: HPT_RENDER_TIME_KERNELS creates HIP events; not available here");
    hpt_context* ctx = s->ctx;
    const uint64_t npix = (uint64_t)s->ds.cam.w * s->ds.cam.h;
    if (npix > HPT_NOALLOC_MAX_PATHS)
        return fail(HPT_ERR_LIMIT, "hpt_render_device_noalloc: frame of %llu pixels exceeds the static workspace (at most %u pixels, e.g. 800 x 600); use hpt_render_device",
                    (unsigned long long)npix, (unsigned)HPT_NOALLOC_MAX_PATHS);
    if (p->spp > HPT_NOALLOC_MAX_SPP)
        return fail(HPT_ERR_LIMIT, "hpt_render_device_noalloc: %d samples per call exceed the static workspace (at most %d; split the range with first_sample)", p->spp, HPT_NOALLOC_MAX_SPP);
    if (!ctx->scratch || ctx->scratch_bytes < (size_t)HPT_NOALLOC_SCRATCH_BYTES) return fail(HPT_ERR_STATE, "hpt_render_device_noalloc: the static workspace of this device was not resolved");
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const double t0 = now_s();
    if (stats) std::memset(stats, 0, sizeof *stats);
    unsigned char* base = (unsigned char*)ctx->scratch;
    DPass ps; std::memset(&ps, 0, sizeof ps);
    for (int a = 0; a < 2; ++a) {
        ps.q[a].ray_o = (float4*)(base + HPT_NOALLOC_OFF_Q(3 * a + 0));
        ps.q[a].ray_d = (float4*)(base + HPT_NOALLOC_OFF_Q(3 * a + 1));
        ps.q[a].thr = (float4*)(base + HPT_NOALLOC_OFF_Q(3 * a + 2));
    }
    ps.hit = (float4*)(base + HPT_NOALLOC_OFF_HIT); ps.radiance = (float4*)(base + HPT_NOALLOC_OFF_RADIANCE);
    ps.hit_inst = (uint32_t*)(base + HPT_NOALLOC_OFF_HIT_INST); ps.mq = (uint4*)(base + HPT_NOALLOC_OFF_MQ); ps.mq_bin = nullptr; ps.counts = (uint32_t*)(base + HPT_NOALLOC_OFF_COUNTS);
    ps.stat = (unsigned long long*)(base + HPT_NOALLOC_OFF_STAT);
    ps.hit_par = s->fuse_analytic ? (uint32_t)HPT_NOALLOC_SLOTS : 0u;
    float2* d_halton = (float2*)(base + HPT_NOALLOC_OFF_HALTON);
    long long* d_accum = (long long*)(base + HPT_NOALLOC_OFF_ACCUM);
    // the array keeps the last call's contents: statistics and the accumulator start at zero here, counters and radiance per pass (run_pass)
    HIPCHK(hipMemsetAsync(ps.stat, 0, 8 * sizeof(unsigned long long), st));
    HIPCHK(hipMemsetAsync(d_accum, 0, npix * sizeof(hpt_accum), st));
    hptk_halton(st, d_halton, (uint32_t)p->first_sample, (uint32_t)p->spp);      // the whole call's samples (<= HPT_NOALLOC_MAX_SPP), once
    // as many samples of every pixel per pass as the workspace holds (hpt_render_device: 64 M paths per pass)
    int S = (int)std::min<uint64_t>(std::min<uint64_t>(HPT_NOALLOC_MAX_PATHS / npix, 64), (uint64_t)p->spp);
    if (p->spp_per_pass > 0) S = std::min(S, p->spp_per_pass);
    EventLog ev{ctx, false};
    uint32_t passes = 0;
    for (int done = 0; done < p->spp; done += S) {
        const int cur = std::min(S, p->spp - done);
        ps.halton = d_halton + done;
        ps.pixel_list = nullptr; ps.n_slots = (uint32_t)npix; ps.S = (uint32_t)cur;
        ps.first_sample = (uint32_t)(p->first_sample + done); ps.seed = p->seed; ps.max_depth = p->max_depth; ps.flags = p->flags;
        ps.shard_cap = shard_cap_for(npix * (uint64_t)cur);
        if ((uint64_t)ps.shard_cap * HPT_SHARDS > HPT_NOALLOC_SLOTS) return fail(HPT_ERR_LIMIT, "hpt_render_device_noalloc: pass of %llu paths does not fit the static workspace", (unsigned long long)(npix * (uint64_t)cur));
        CHK(run_pass(s, ps, d_accum, nullptr, (p->flags & HPT_RENDER_COUNTERS) != 0, ev, extend_grid(s, 1), st, nullptr, nullptr));
        ++passes;
    }
    hptk_accum_resolve(st, ctx->max_blocks, d_accum, nullptr, rgb_sum_device, (uint32_t)npix, 0.0f);
    HIPCHK(hipGetLastError());
    if (stats) { stats->passes = passes; stats->samples = npix * (uint64_t)p->spp; }
    return finish_stats(s, stats, ev, t0, ps.stat);
}

extern "C" int hpt_render_samples(hpt_scene* s, const hpt_render_params* p, float* per_sample) {
    CHK(check_params(s, p));
    if (p->adaptive || !per_sample) return fail(HPT_ERR_INVALID, "hpt_render_samples: fixed-spp only, non-null output");
    hpt_context* ctx = s->ctx;
    HIPCHK(hipSetDevice(ctx->device));
    const uint64_t npix = (uint64_t)s->ds.cam.w * s->ds.cam.h;
    CHK(s->w_accum.ensure(npix * sizeof(hpt_accum)));
    CHK(s->w_stat.ensure(8 * sizeof(unsigned long long)));
    HIPCHK(hipMemsetAsync(s->w_stat.p, 0, 8 * sizeof(unsigned long long), ctx->stream));
    HIPCHK(hipMemsetAsync(s->w_accum.p, 0, npix * sizeof(hpt_accum), ctx->stream));
    EventLog ev{ctx, false};
    CHK(render_fixed(s, p, s->w_accum.as<long long>(), per_sample, nullptr, ev));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return HPT_OK;
}

extern "C" int hpt_render(hpt_scene* s, const hpt_render_params* p, float* rgb, uint32_t* samples_per_px, hpt_render_stats* stats) {
    CHK(check_params(s, p));
    if (!rgb) return fail(HPT_ERR_INVALID, "hpt_render: null output");
    hpt_context* ctx = s->ctx;
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const double t0 = now_s();
    if (stats) std::memset(stats, 0, sizeof *stats);
    const uint64_t npix = (uint64_t)s->ds.cam.w * s->ds.cam.h;
    CHK(s->w_accum.ensure(npix * sizeof(hpt_accum))); CHK(s->w_out.ensure(npix * 3 * sizeof(float)));
    CHK(s->w_stat.ensure(8 * sizeof(unsigned long long)));
    HIPCHK(hipMemsetAsync(s->w_stat.p, 0, 8 * sizeof(unsigned long long), st));
    HIPCHK(hipMemsetAsync(s->w_accum.p, 0, npix * sizeof(hpt_accum), st));
    long long* accum = s->w_accum.as<long long>();
    EventLog ev{ctx, stats && (p->flags & HPT_RENDER_TIME_KERNELS)};
    if (!p->adaptive) {
        CHK(render_fixed(s, p, accum, nullptr, stats, ev));
        hptk_accum_resolve(st, ctx->max_blocks, accum, nullptr, s->w_out.as<float>(), (uint32_t)npix, (float)p->spp);
        if (samples_per_px) for (uint64_t i = 0; i < npix; ++i) samples_per_px[i] = (uint32_t)p->spp;
    } else {
        // main.rs:67-107 in wavefront form: rounds of samples (have, num_samples] over the still-active pixels
        CHK(s->w_minmax.ensure(npix * 6 * sizeof(float))); CHK(s->w_spp.ensure(npix * sizeof(uint32_t)));
        CHK(s->w_list[0].ensure(npix * sizeof(uint32_t))); CHK(s->w_list[1].ensure(npix * sizeof(uint32_t)));
        CHK(s->w_listcount.ensure(sizeof(uint32_t)));
        // min = +inf, max = 0 (main.rs:90-92)
        hptk_init_minmax(st, ctx->max_blocks, s->w_minmax.as<float>(), (uint32_t)npix);
        hptk_fill_u32(st, ctx->max_blocks, s->w_spp.as<uint32_t>(), npix, (uint32_t)p->initial_samples);
        int have = 0, num_samples = p->initial_samples, cur_list = -1;   // -1: all pixels
        uint32_t n_active = (uint32_t)npix;
        while (n_active > 0) {
            hpt_render_params rp = *p;   // this round: samples (have, num_samples] of the still-active pixels
            rp.adaptive = 0; rp.first_sample = have; rp.spp = num_samples - have;
            CHK(render_range(s, &rp, cur_list < 0 ? nullptr : s->w_list[cur_list].as<uint32_t>(), n_active, accum, s->w_minmax.as<float>(), nullptr, stats, ev));
            have = num_samples;
            if (num_samples >= p->max_samples) break;   // main.rs:101 first clause
            const int next = cur_list < 0 ? 0 : 1 - cur_list;
            HIPCHK(hipMemsetAsync(s->w_listcount.p, 0, sizeof(uint32_t), st));
            hptk_adaptive_select(st, ctx->max_blocks, cur_list < 0 ? nullptr : s->w_list[cur_list].as<uint32_t>(), n_active, s->w_minmax.as<float>(),
                                 p->thresholds, s->w_list[next].as<uint32_t>(), s->w_listcount.as<uint32_t>(), s->w_spp.as<uint32_t>(), (uint32_t)(num_samples * 2));
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(st));
            HIPCHK(hipMemcpy(&n_active, s->w_listcount.p, sizeof(uint32_t), hipMemcpyDeviceToHost));
            cur_list = next;
            num_samples *= 2;
        }
        hptk_accum_resolve(st, ctx->max_blocks, accum, s->w_spp.as<uint32_t>(), s->w_out.as<float>(), (uint32_t)npix, 0.0f);
        HIPCHK(hipStreamSynchronize(st));
        if (samples_per_px) HIPCHK(hipMemcpy(samples_per_px, s->w_spp.p, npix * sizeof(uint32_t), hipMemcpyDeviceToHost));
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipMemcpy(rgb, s->w_out.p, npix * 3 * sizeof(float), hipMemcpyDeviceToHost));
    if (stats) return finish_stats(s, stats, ev, t0);
    return HPT_OK;
}

// Samples first_sample+1 ..= first_sample+spp of the listed pixels ADDED to a caller-held accumulator on the DEVICE: the
// building block of everything that renders a frame in several calls -- a rank of a multi-GPU render (its share of the
// samples; the accumulators of the ranks are then summed, ncclInt64), one round of the adaptive loop driven by the host
// (hpt/dist.py, hpt_group.cpp), a progressive or resumed render.  The sums are integers: the split changes no bit.
extern "C" int hpt_render_accum_device(hpt_scene* s, const hpt_render_params* p, const uint32_t* pixel_list_device, uint32_t n_pixels,
                                       hpt_accum* accum_device, float* minmax_device, hpt_render_stats* stats) {
    CHK(check_params(s, p));
    if (p->adaptive) return fail(HPT_ERR_INVALID, "hpt_render_accum_device: params describe ONE sample range (adaptive = 0, first_sample, spp)");
    if (!accum_device) return fail(HPT_ERR_INVALID, "hpt_render_accum_device: null accumulator");
    const uint64_t npix = (uint64_t)s->ds.cam.w * s->ds.cam.h;
    if (!pixel_list_device) n_pixels = (uint32_t)npix;
    if (n_pixels > npix) return fail(HPT_ERR_INVALID, "hpt_render_accum_device: %u listed pixels, image has %llu", n_pixels, (unsigned long long)npix);
    hpt_context* ctx = s->ctx;
    HIPCHK(hipSetDevice(ctx->device));
    const double t0 = now_s();
    if (stats) std::memset(stats, 0, sizeof *stats);
    CHK(s->w_stat.ensure(8 * sizeof(unsigned long long)));
    HIPCHK(hipMemsetAsync(s->w_stat.p, 0, 8 * sizeof(unsigned long long), ctx->stream));
    EventLog ev{ctx, stats && (p->flags & HPT_RENDER_TIME_KERNELS)};
    CHK(render_range(s, p, pixel_list_device, n_pixels, (long long*)accum_device, minmax_device, nullptr, stats, ev));
    if (stats) { stats->samples = (uint64_t)n_pixels * (uint64_t)p->spp; }
    return finish_stats(s, stats, ev, t0);
}

// The same with the accumulator in HOST memory (uploaded, added to, downloaded): resumable rendering for a host that holds
// no device memory of its own -- the command-line programs (host/main.cpp --checkpoint / --resume, rust/main_render.rs).
extern "C" int hpt_render_accum(hpt_scene* s, const hpt_render_params* p, hpt_accum* accum, hpt_render_stats* stats) {
    CHK(check_params(s, p));
    if (p->adaptive) return fail(HPT_ERR_INVALID, "hpt_render_accum: fixed sample range only (adaptive = 0, first_sample, spp)");
    if (!accum) return fail(HPT_ERR_INVALID, "hpt_render_accum: null accumulator");
    hpt_context* ctx = s->ctx;
    HIPCHK(hipSetDevice(ctx->device));
    const uint64_t npix = (uint64_t)s->ds.cam.w * s->ds.cam.h;
    CHK(s->w_accum.ensure(npix * sizeof(hpt_accum)));
    HIPCHK(hipMemcpyAsync(s->w_accum.p, accum, npix * sizeof(hpt_accum), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));      // (pageable host memory)
    CHK(hpt_render_accum_device(s, p, nullptr, 0, s->w_accum.as<hpt_accum>(), nullptr, stats));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipMemcpy(accum, s->w_accum.p, npix * sizeof(hpt_accum), hipMemcpyDeviceToHost));
    return HPT_OK;
}

// accumulator -> image (main.rs:109-110: total / num_samples as f32), device and host form of the same arithmetic
extern "C" int hpt_accum_resolve_device(hpt_scene* s, const hpt_accum* accum_device, const uint32_t* samples_per_px_device, uint32_t num_samples, float* rgb_device) {
    CHK(need_committed(s));
    if (!accum_device || !rgb_device) return fail(HPT_ERR_INVALID, "hpt_accum_resolve_device: null pointer");
    hpt_context* ctx = s->ctx;
    HIPCHK(hipSetDevice(ctx->device));
    hptk_accum_resolve(ctx->stream, ctx->max_blocks, (const long long*)accum_device, samples_per_px_device, rgb_device, (uint32_t)((uint64_t)s->ds.cam.w * s->ds.cam.h), (float)num_samples);
    HIPCHK(hipGetLastError());
    return HPT_OK;
}
extern "C" int hpt_accum_resolve(const hpt_accum* accum, uint32_t n_pixels, const uint32_t* samples_per_px, uint32_t num_samples, float* rgb) {
    if (!accum || !rgb) return fail(HPT_ERR_INVALID, "hpt_accum_resolve: null pointer");
    for (uint32_t i = 0; i < n_pixels; ++i) {
        const hpt_accum& a = accum[i];
        const float n = samples_per_px ? (float)samples_per_px[i] : (float)num_samples;
        float v[3] = {(float)((double)a.r * (1.0 / HPT_ACCUM_SCALE_D)), (float)((double)a.g * (1.0 / HPT_ACCUM_SCALE_D)), (float)((double)a.b * (1.0 / HPT_ACCUM_SCALE_D))};
        for (int c = 0; c < 3; ++c) {
            if (a.nan != 0) v[c] = std::nanf("");
            if (samples_per_px || num_samples != 0) v[c] = v[c] / n;
            rgb[3 * (size_t)i + c] = v[c];
        }
    }
    return HPT_OK;
}

// main.rs:101-106 on DEVICE buffers: which of the listed pixels go on to the next round.
extern "C" int hpt_adaptive_select_device(hpt_scene* s, const uint32_t* list_in_device, uint32_t n_in, const float* minmax_device,
                                          const float thresholds[3], uint32_t* list_out_device, uint32_t* samples_per_px_device,
                                          uint32_t next_num_samples, uint32_t* n_out) {
    CHK(need_committed(s));
    if (!minmax_device || !thresholds || !list_out_device || !samples_per_px_device || !n_out) return fail(HPT_ERR_INVALID, "hpt_adaptive_select_device: null argument");
    const uint64_t npix = (uint64_t)s->ds.cam.w * s->ds.cam.h;
    if (!list_in_device) n_in = (uint32_t)npix;
    if (n_in > npix) return fail(HPT_ERR_INVALID, "hpt_adaptive_select_device: %u listed pixels, image has %llu", n_in, (unsigned long long)npix);
    hpt_context* ctx = s->ctx;
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    *n_out = 0;
    if (n_in == 0) return HPT_OK;
    CHK(s->w_listcount.ensure(sizeof(uint32_t)));
    HIPCHK(hipMemsetAsync(s->w_listcount.p, 0, sizeof(uint32_t), st));
    hptk_adaptive_select(st, ctx->max_blocks, list_in_device, n_in, minmax_device, thresholds, list_out_device, s->w_listcount.as<uint32_t>(),
                         samples_per_px_device, next_num_samples);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipMemcpy(n_out, s->w_listcount.p, sizeof(uint32_t), hipMemcpyDeviceToHost));
    return HPT_OK;
}

extern "C" int hpt_scene_info(hpt_scene* s, hpt_scene_info_t* out) {
    CHK(need_committed(s));
    if (!out) return fail(HPT_ERR_INVALID, "hpt_scene_info: null output");
    std::memset(out, 0, sizeof *out);
    out->n_instances = (int32_t)s->ds.n_instances; out->n_mesh_instances = (int32_t)s->ds.n_mesh_instances;
    out->n_meshes = (int32_t)s->meshes.size(); out->n_materials = s->n_mat;
    out->n_nodes = s->ds.n_nodes; out->n_triangles = s->n_tris_total; out->tlas_nodes = s->tlas_nodes;
    out->uses_tlas = s->ds.use_tlas; out->stack_entries = s->stack_entries; out->device_bytes = s->device_bytes;
    out->node_bytes = s->node_bytes; out->trees_optimized = s->trees_optimized;
    return HPT_OK;
}

// ------------------------------------------------------------------------------------------------ internal seams (hpt_internal.h)
int hpti_fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_err = buf;
    return code;
}
uint64_t hpti_scene_generation(const hpt_scene* s) { return s->generation; }
hpt_context* hpti_scene_context(const hpt_scene* s) { return s->ctx; }
void hpti_scene_frame(const hpt_scene* s, uint32_t* w, uint32_t* h) { *w = s->camera.img_width; *h = s->camera.img_height; }
hipStream_t hpti_context_stream(const hpt_context* c) { return c->stream; }
int hpti_context_device(const hpt_context* c) { return c->device; }
int hpti_context_max_blocks(const hpt_context* c) { return c->max_blocks; }
int hpti_scene_clone(const hpt_scene* src, hpt_context* ctx, hpt_scene** out) {
    if (!src || !ctx || !out) return fail(HPT_ERR_INVALID, "scene clone: null argument");
    if (!src->committed) return fail(HPT_ERR_STATE, "scene not committed: call hpt_scene_commit first");
    hpt_scene* s = nullptr;
    CHK(hpt_scene_create(ctx, &s));
    s->meshes = src->meshes; s->nodes = src->nodes; s->top = src->top; s->materials = src->materials; s->images = src->images;
    s->background = src->background; s->environment = src->environment; s->camera = src->camera; s->have_camera = src->have_camera;
    const int rc = hpt_scene_commit(s);
    if (rc != HPT_OK) { const std::string keep = g_err; hpt_scene_destroy(s); g_err = keep; return rc; }
    *out = s;
    return HPT_OK;
}

// ------------------------------------------------------------------------------------------------ tuning hooks
// Not part of include/hpt.h: used by scripts/gpu_time.py for interleaved A/B runs inside one process.
extern "C" void hpt_debug_set_tree_opt(int mode) { g_tree_opt_override = mode; }      // -1: environment / default; applies to later commits
extern "C" void hpt_debug_set_refill(int lanes) { hptk_set_refill(lanes); }
extern "C" void hpt_debug_set_overlap(hpt_context* ctx, int n) { if (ctx) ctx->overlap = std::min(std::max(n, 1), HPT_MAX_INFLIGHT); }
extern "C" void hpt_debug_set_extend_blocks(hpt_context* ctx, int per_cu) { if (ctx && per_cu >= 0 && per_cu <= 16) ctx->extend_per_cu = per_cu; }
