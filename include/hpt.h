/*
 * hpt.h -- C ABI of the MI355X-native render core for glowcoil/path-tracer (libhpt.so).
 *
 * The reference (Rust, /root/reference) has no FFI or plugin interface; the seam is the one BASELINE.json's
 * north_star names: the host keeps load.rs (load_scene, load.rs:17) and the CLI (main.rs:34-41), and replaces the
 * buffer allocation + pixel loop of main.rs:55-115 -- i.e. Scene::sample (main.rs:85 -> scene.rs:126) and everything
 * under it -- with calls into this library, then runs save_img (main.rs:117).  Each entry point cites what it replaces.
 *
 * Conventions
 *  - every function returns 0 on success or a negative hpt_status; nothing unwinds across the boundary; the text
 *    of the last error on the calling thread is hpt_last_error().  (The reference panics, exit 101, on every error.)
 *  - plain pointers and sizes only.  The library copies every input array during the call (caller keeps ownership).
 *  - matrices are 3x3 COLUMN-major float[9] (m[c*3+r]) like cgmath::Matrix3 (scene.rs:102-105 Transform).
 *  - one hpt_context drives ONE GPU from one host thread.  Several GPUs: either one process per GPU, each with its own
 *    context adding a sample range (first_sample, spp) of every pixel to an accumulator (hpt_render_accum_device), the
 *    accumulators summed afterwards (bench.py: torch.distributed over RCCL, int64 sum), or ONE process and an hpt_group
 *    (below), which does the split and the RCCL reduce itself.  The accumulator is integer (hpt_accum): the image does
 *    not depend, in any bit, on how many GPUs rendered it.
 *  - destroy scenes before their context (a context destroyed earlier is kept alive until its last scene goes).
 *  - a context is not thread-safe.  All work of a context is ordered on one HIP stream (its own, or the one
 *    given to hpt_context_set_stream): it starts behind what that stream holds and is complete when that stream
 *    reaches the end of the call's work (a render runs every other pass on a second stream of the library's own,
 *    forked from and joined back into the context stream); host-pointer entry points synchronise that stream before
 *    returning, *_device entry points do not.
 */
#ifndef HPT_H
#define HPT_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct hpt_context hpt_context;
typedef struct hpt_scene hpt_scene;

typedef enum {
    HPT_OK = 0,
    HPT_ERR_INVALID = -1,   /* bad argument / malformed input (reference: panic!/expect/unwrap) */
    HPT_ERR_DEVICE = -2,    /* HIP runtime error */
    HPT_ERR_NOMEM = -3,
    HPT_ERR_STATE = -4,     /* call order (e.g. render before commit) */
    HPT_ERR_LIMIT = -5      /* size beyond a documented limit */
} hpt_status;

enum { HPT_GEOM_NONE = 0, HPT_GEOM_SPHERE = 1, HPT_GEOM_PLANE = 2, HPT_GEOM_MESH = 3 };   /* geometry.rs:11-15 (NONE: Node.object == None) */
enum { HPT_TEX_BLANK = 0, HPT_TEX_CHECKER = 1, HPT_TEX_IMAGE = 2 };                        /* scene.rs:54-58 TextureData */

/* bvh.rs:13-23 BVHNode in the pre-order Vec the host built (bvh.rs:35-146).
 * interior: left/right = child indices, prim = -1;  leaf: left = right = -1, prim = primitive index. */
typedef struct {
    int32_t left, right, prim;
    float p1[3], p2[3];
} hpt_bvh_node;

/* scene.rs:46-58 Texture */
typedef struct {
    int32_t kind;            /* HPT_TEX_* */
    int32_t image;           /* id from hpt_scene_add_image for HPT_TEX_IMAGE */
    float color[3];
    float color1[3], color2[3];
    float m[9], t[3];        /* texture transform */
} hpt_texture;

/* scene.rs:33-44 Material -- only the fields Scene::cast reads (specular, glossiness, absorption are dead data) */
typedef struct {
    hpt_texture diffuse, reflection, refraction;
    float emission[3];
    float reflection_glossiness, refraction_glossiness, refraction_index;
} hpt_material;

/* scene.rs:76-85 Camera exactly as load_camera (load.rs:241-268) leaves it: dir normalised, up re-orthogonalised
 * and NOT renormalised. */
typedef struct {
    float pos[3], dir[3], up[3];
    float fov;               /* vertical, degrees */
    uint32_t img_width, img_height;
    float focaldist, dof;
} hpt_camera;

enum {
    HPT_RENDER_COUNTERS = 1u,      /* run the instrumented extend kernel: fills nodes_visited / tris_tested / inst_tests */
    HPT_RENDER_TIME_KERNELS = 2u,  /* bracket every kernel with HIP events on the context stream; fills seconds_*.  Runs the
                                      passes one after the other on that stream (without the flag a render keeps two
                                      wavefront passes in flight, the second on a stream of the library's own, joined
                                      back into the context stream before the call's work ends; same results bit for bit) */
    HPT_RENDER_EXACT_SHADE = 4u    /* shade with the reference's own f32 operations (IEEE divide and square root, libm-style
                                      sincosf / cosf(asinf z): scene.rs:211-225, 394-429) instead of the default 1-ulp forms
                                      (v_rcp / v_rsq, a quadrant polynomial for sin/cos of 2 pi r, sqrt(1 - z^2)).  Hits are
                                      decided by exact arithmetic either way; this flag only concerns the bounce.  The default
                                      differs from it by a few ulp per bounce direction -- inside the image-RMSE contract, and
                                      what the ~0.003 % of paths whose roulette / lobe / hit decision sits within that distance
                                      do differently.  With the flag, path lengths equal the CPU oracle's one for one (the
                                      parity tests use it); the shade kernel is ~10 % slower. */
};

/* replaces the constants of main.rs:27-32 and the loop bounds of main.rs:67-107 */
typedef struct {
    int32_t adaptive;         /* 1: reference behaviour (initial_samples, doubling up to max_samples); 0: fixed spp */
    int32_t spp;              /* fixed mode: samples per pixel rendered by THIS call */
    int32_t first_sample;     /* fixed mode: samples first_sample+1 ..= first_sample+spp (Halton index, main.rs:72) */
    int32_t initial_samples;  /* main.rs:31 */
    int32_t max_samples;      /* main.rs:32 */
    float thresholds[3];      /* main.rs:27-29 */
    int32_t max_depth;        /* bounces after the primary hit; a hit at depth max_depth adds its emission only.
                                 Deliberate deviation: the reference recursion is unbounded. */
    uint32_t seed;
    int32_t spp_per_pass;     /* samples of every pixel in flight per wavefront pass; 0 = library default */
    uint32_t flags;           /* HPT_RENDER_* */
} hpt_render_params;

typedef struct {
    uint64_t rays_primary, rays_bounce;      /* Scene::intersect-equivalent segments traced */
    uint64_t nodes_visited, tris_tested, inst_tests;   /* HPT_RENDER_COUNTERS */
    double seconds_total;                    /* host wall clock around the whole call (after final sync) */
    double seconds_raygen, seconds_extend, seconds_shade, seconds_resolve;   /* HPT_RENDER_TIME_KERNELS (HIP events) */
    uint32_t launches_extend, passes;
    uint64_t samples;                        /* pixel-samples rendered */
    double seconds_analytic;                 /* HPT_RENDER_TIME_KERNELS: the analytic stage of the extend step (k_extend_a: spheres, planes, mesh
                                                root boxes; instance-list scenes) -- seconds_extend is the BVH traversal kernel alone */
    uint64_t rays_mesh;                      /* rays that entered a mesh box = the rays the traversal kernel walked (instance-list scenes) */
} hpt_render_stats;

typedef struct {
    uint64_t rays, nodes_visited, tris_tested;
    double seconds_kernel;     /* HIP-event time of the traversal kernel alone */
    double seconds_total;      /* incl. H2D/D2H copies for the host-pointer entry */
} hpt_trace_stats;

/* The radiance accumulator of one pixel: main.rs:67,109 collect a pixel's samples and add them up (f32, in sample order).
 * Here the sums are 64-bit FIXED-POINT numbers -- sample * 2^HPT_ACCUM_FRAC_BITS rounded to nearest, |sample| clamped to
 * HPT_ACCUM_SAMPLE_MAX first (+-inf included); a sample with a NaN channel adds 0 to that channel and 1 to `nan` -- because
 * integer addition is associative: the accumulator holds the same bits whatever the pass size, the number of GPUs the
 * samples were split over, the order in which RCCL adds the ranks, or the points at which a render was stopped and resumed.
 * The rounding error is <= 2^-25 per sample (the reference's own f32 fold is coarser); sums stay below 2^63 for up to
 * 2^18 samples per pixel even if every one of them is at the clamp.  hpt_accum_resolve turns it into the image. */
#define HPT_ACCUM_FRAC_BITS 24
#define HPT_ACCUM_SAMPLE_MAX 1048576.0f
typedef struct {
    int64_t r, g, b;           /* fixed-point sums of the samples' radiance */
    int64_t nan;               /* samples that had a NaN channel: the pixel resolves to NaN, as the reference's f32 sum would */
} hpt_accum;

typedef struct {               /* what hpt_scene_commit laid out in HBM */
    int32_t n_instances, n_mesh_instances, n_meshes, n_materials;
    int32_t n_nodes;           /* Node64 records, all BLASes + the TLAS */
    int32_t tlas_nodes;        /* of which TLAS (0 when the instance list is walked linearly, scene.rs:193) */
    int32_t uses_tlas, stack_entries;
    int64_t n_triangles;
    uint64_t device_bytes;     /* scene data resident in HBM (render workspaces excluded) */
    int32_t node_bytes;        /* 32: 16-bit child boxes (default when the grid resolves the leaves), 64: float boxes */
    int32_t trees_optimized;   /* meshes whose tree hpt_scene_commit restructured (SAH re-partition / insertion rounds) before the layout */
} hpt_scene_info_t;

/* ---- misc ---- */
const char* hpt_version(void);
const char* hpt_last_error(void);
int hpt_device_count(void);
/* Counter-based generator that replaces rand::random::<f32>() (main.rs:76-77, scene.rs:161,169,407,408,425,426).
 *   mix(x): x^=x>>16; x*=0x7feb352d; x^=x>>15; x*=0x846ca68b; x^=x>>16
 *   k1 = mix(pixel + mix(seed));  k2 = mix(sample ^ 0x5bd1e995 ^ seed*0x9E3779B9)
 *   u  = (mix(k1 ^ (k2 + dim*0x9E3779B9)) >> 8) * 2^-24                in [0,1)   (one hash per number on top of the two key hashes)
 * pixel = y*W+x, sample = Halton index (1-based), dim: 0,1 lens; 2+4*depth+{0 roulette,1 lobe,2 z,3 theta}. */
float hpt_rng_f32(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t dim);
float hpt_halton(int32_t index, int32_t base);                                   /* scene.rs:431 */
int hpt_tonemap_rgba8(const float* rgb, uint32_t n_pixels, uint8_t* rgba);       /* scene.rs:362 color_as_u8_array */

/* ---- context ---- */
int hpt_context_create(int device, hpt_context** out);
void hpt_context_destroy(hpt_context* ctx);
/* use the caller's hipStream_t (e.g. torch's current stream) for all later work; NULL = back to the context's own */
int hpt_context_set_stream(hpt_context* ctx, void* hip_stream);
int hpt_context_synchronize(hpt_context* ctx);
/* wavefront passes a render keeps in flight: 1 = everything on the context stream, 2 (default) = every other pass on a
 * second stream of the library's own with a second set of pass buffers (+8.9 GB at the default pass size), up to 4.
 * No counterpart in the reference (main.rs:57 is one rayon loop); results do not depend on it. */
int hpt_context_set_passes_in_flight(hpt_context* ctx, int n);

/* ---- scene building: what load_scene (load.rs:17) returns, handed over piece by piece ---- */
int hpt_scene_create(hpt_context* ctx, hpt_scene** out);
void hpt_scene_destroy(hpt_scene* scene);
/* geometry.rs:18-27 Mesh as load_obj (load.rs:282-321) builds it: 0-based index triples (ntri x 3) for position,
 * normal and texture vertex; uvw is nt x 3.  nodes/nnodes = the BVH.nodes Vec that Mesh::build_bvh
 * (geometry.rs:223, load.rs:310) produced on the host; the library validates it and, at hpt_scene_commit, restructures
 * it over the same leaves (SAH re-partition + insertion rounds: bvh.rs:48-146 falls back to the index median on all
 * but a few splits), refits tight boxes and flattens it to the device layout.  nodes == NULL: the library builds its
 * own SAH tree instead.  Hits do not depend on the tree: images are bit-identical either way. */
int hpt_scene_add_mesh(hpt_scene* scene, const float* verts, int32_t nv, const uint32_t* tri_idx,
                       const float* normals, int32_t nn, const uint32_t* nrm_idx,
                       const float* uvw, int32_t nt, const uint32_t* uv_idx, int32_t ntri,
                       const hpt_bvh_node* nodes, int32_t nnodes, int32_t* mesh_id);
/* TextureData::Image (scene.rs:56, load_img load.rs:437): RGB8, 3 bytes per pixel, row-major */
int hpt_scene_add_image(hpt_scene* scene, const uint8_t* rgb8, int32_t width, int32_t height, int32_t* image_id);
/* one entry of Scene.materials (scene.rs:12); objects refer to it by the returned id instead of by name */
int hpt_scene_add_material(hpt_scene* scene, const hpt_material* material, int32_t* material_id);
/* scene.rs:19-24 Node.  parent = -1: an element of Scene.nodes; otherwise a child of that node.  Call in the
 * reference's own order (parents first, children in Vec order): tie-breaking follows it (scene.rs:196,251). */
int hpt_scene_add_node(hpt_scene* scene, int32_t parent, int32_t geom, int32_t mesh_id, int32_t material_id,
                       const float m[9], const float t[3], int32_t* node_id);
int hpt_scene_set_background(hpt_scene* scene, const hpt_texture* tex);     /* Scene.background  (scene.rs:14) */
int hpt_scene_set_environment(hpt_scene* scene, const hpt_texture* tex);    /* Scene.environment (scene.rs:15) */
int hpt_scene_set_camera(hpt_scene* scene, const hpt_camera* cam);          /* Camera (scene.rs:76) */
/* flatten + upload (replicated per GPU: every rank commits the same scene) */
int hpt_scene_commit(hpt_scene* scene);
int hpt_scene_info(hpt_scene* scene, hpt_scene_info_t* out);   /* no counterpart in the reference: sizes for logs and roofline accounting */

/* ---- parity entry: Mesh::intersect (geometry.rs:132-158) over n local-space rays against one mesh ----
 * prim = -1 on miss (t,u,v = 0); side: 1 = Side::Front, 0 = Side::Back.  Host pointers; outputs caller-allocated. */
int hpt_trace_rays(hpt_scene* scene, int32_t mesh_id, int64_t n, const float* org3, const float* dir3,
                   int32_t* prim, float* t, float* u, float* v, uint8_t* side, hpt_trace_stats* stats);
/* same with DEVICE pointers (org4/dir4: n x float4, .w ignored; hit4: n x {t,u,v,prim-as-int-bits}, side in bit 31
 * of inst word is not used here: side_out n x uint8).  Enqueues on the context stream and returns; with
 * stats != NULL it synchronises and reports the HIP-event kernel time. count_visits = 1 runs the instrumented kernel. */
int hpt_trace_rays_device(hpt_scene* scene, int32_t mesh_id, int64_t n, const void* org4, const void* dir4,
                          void* hit4, void* side_out, int count_visits, hpt_trace_stats* stats);

/* ---- test entry: Scene::intersect (scene.rs:190-208) over n world-space rays ----
 * node = id from hpt_scene_add_node (-1 miss); pos/normal/uvw = the HitInfo after every from_local_space step. */
int hpt_scene_intersect(hpt_scene* scene, int64_t n, const float* org3, const float* dir3,
                        int32_t* node, int32_t* prim, float* t, uint8_t* side,
                        float* pos3, float* normal3, float* uvw3);

/* ---- render: main.rs:57-110 minus the tonemap ----
 * rgb: W*H*3 floats, caller-allocated HOST memory; receives sum/num_samples per pixel (main.rs:109-110).
 * samples_per_px (optional, W*H): the num_samples each pixel ended with. */
int hpt_render(hpt_scene* scene, const hpt_render_params* params, float* rgb, uint32_t* samples_per_px,
               hpt_render_stats* stats);
/* fixed-spp only, DEVICE output: rgb_sum (W*H*3 floats, device memory) receives the per-pixel SUM over this
 * call's sample range (no division), i.e. the accumulator's sums rounded to f32.  Synchronises the context stream
 * before returning only if stats != NULL.  (To combine several calls -- ranks, rounds, a resumed render -- without a
 * dependence on the split, add into ONE accumulator with hpt_render_accum_device instead of adding these floats.) */
int hpt_render_device(hpt_scene* scene, const hpt_render_params* params, float* rgb_sum_device,
                      hpt_render_stats* stats);
/* hpt_render_device for callers that cannot allocate -- tight serving loops, harnesses that forbid hipMalloc.  No
 * counterpart in the reference (its main.rs:55 allocates the frame buffer per run).  Same arguments, same kernels and
 * bit-identical output as hpt_render_device, but the pass buffers (wavefront queues, hit and radiance records, counters,
 * Halton table, the call's accumulator) come from a static device array of HPT_NOALLOC_SCRATCH_MB MB that each GPU the
 * library runs on holds until the process ends, instead of from hipMalloc: the call allocates nothing (no hipMalloc /
 * hipFree, no stream or event creation, no host allocation) and copies nothing from the host; with stats == NULL it does
 * not synchronise either -- it enqueues kernels and memsets on the context stream and returns (stats != NULL: one
 * synchronisation at the end, to read the counters).  It has not been tried under HIP graph capture.  Limits
 * (HPT_ERR_LIMIT beyond them): the frame has at most HPT_NOALLOC_MAX_PIXELS = 2^19 pixels (the reference's 800 x 600
 * fits) and a call renders at most HPT_NOALLOC_MAX_SPP = 4096 samples (continue with first_sample).  Fixed-spp only;
 * HPT_RENDER_TIME_KERNELS is refused (it creates events).  One pass at a time on the context stream.  Every call on a
 * GPU shares that GPU's one array (whatever the context or scene): calls that overlap on one GPU race -- serialise them. */
#define HPT_NOALLOC_MAX_PIXELS (1u << 19)
#define HPT_NOALLOC_MAX_SPP 4096
#define HPT_NOALLOC_SCRATCH_MB 127
int hpt_render_device_noalloc(hpt_scene* scene, const hpt_render_params* params, float* rgb_sum_device,
                              hpt_render_stats* stats);
/* Samples first_sample+1 ..= first_sample+spp (params: adaptive = 0) of the listed pixels ADDED to a caller-held
 * accumulator on the DEVICE (W*H records, zeroed by the caller before the first call): main.rs:72-99 restricted to a
 * sample range.  The building block of every render made of several calls -- a rank of a multi-GPU render adds its share
 * of the samples and the ranks' accumulators are summed (int64: ncclInt64 / torch.int64); a host that drives the
 * adaptive rounds itself (min/max planes all-reduced between rounds); a progressive or resumed render -- and
 * none of these splits changes a bit of the result.  pixel_list (device, n_pixels entries; NULL = every pixel in order)
 * = the pixels still active; minmax (W*H*6: per pixel min r,g,b then max r,g,b; the caller initialises min = +inf,
 * max = 0 as main.rs:90-92) is folded into; NULL skips it.  Synchronises only if stats != NULL. */
int hpt_render_accum_device(hpt_scene* scene, const hpt_render_params* params, const uint32_t* pixel_list_device,
                            uint32_t n_pixels, hpt_accum* accum_device, float* minmax_device, hpt_render_stats* stats);
/* The same for every pixel with the accumulator in HOST memory (copied in, added to, copied out): resumable rendering for
 * hosts that hold no device memory -- a checkpoint is this array + the sample cursor (host/main.cpp --checkpoint). */
int hpt_render_accum(hpt_scene* scene, const hpt_render_params* params, hpt_accum* accum, hpt_render_stats* stats);
/* accumulator -> image, main.rs:109-110 (`total / num_samples as f32`): rgb = f32(sum) / n with n = samples_per_px[pixel]
 * if given, else num_samples; num_samples == 0 without samples_per_px: no division (the sums).  Device form (context
 * stream, no synchronisation; W*H pixels of the scene's camera) and host form: the same IEEE operations, the same bits. */
int hpt_accum_resolve_device(hpt_scene* scene, const hpt_accum* accum_device, const uint32_t* samples_per_px_device,
                             uint32_t num_samples, float* rgb_device);
int hpt_accum_resolve(const hpt_accum* accum, uint32_t n_pixels, const uint32_t* samples_per_px, uint32_t num_samples, float* rgb);
/* The contrast test of main.rs:101-106 on DEVICE buffers: list_out receives those pixels of list_in (NULL = all)
 * with (max-min)/(min+max) above a threshold in any channel, their samples_per_px entry is set to next_num_samples,
 * *n_out (host) = how many.  The order of list_out is unspecified (results do not depend on it).  Synchronises. */
int hpt_adaptive_select_device(hpt_scene* scene, const uint32_t* list_in_device, uint32_t n_in, const float* minmax_device,
                               const float thresholds[3], uint32_t* list_out_device, uint32_t* samples_per_px_device,
                               uint32_t next_num_samples, uint32_t* n_out);
/* optional per-sample radiance dump for parity tests (fixed mode): per_sample = W*H*spp*3 floats on the HOST,
 * pixel-major then sample. */
int hpt_render_samples(hpt_scene* scene, const hpt_render_params* params, float* per_sample);

/* ---- one call that uses every GPU of the node: main.rs:57-115 (one loop nest that saturates the whole machine) ----
 * A group holds one context per listed GPU (n_devices == 0: every GPU the process sees) and, unless a device is listed
 * twice or HPT_GROUP_TRANSPORT=p2p is set, an RCCL communicator over them (ncclCommInitAll; librccl.so.1 is loaded at
 * this call).  The host builds and commits ONE scene on hpt_group_context(group, 0); hpt_group_render replicates it on
 * the other GPUs (again whenever it was re-committed), renders a contiguous share of the samples of every pixel on each
 * GPU from a host thread of its own, sum-reduces the accumulators onto GPU 0 (ncclReduce of int64 over xGMI; transport
 * p2p: peer copies + adds) and returns sum / num_samples on the host like hpt_render.  The RNG is keyed by (pixel,
 * sample) and the accumulators are integer (hpt_accum): the image is the one-GPU image, bit for bit, for any GPU count.
 * Adaptive mode runs the rounds of main.rs:67-107 in lock step: min/max planes all-reduced per round, identical active
 * sets on every GPU, per-pixel sample counts identical to the one-GPU render.
 * Destroy the scene before the group.  Not thread-safe; one host thread drives a group. */
typedef struct hpt_group hpt_group;
typedef struct {
    int32_t n_devices;
    int32_t transport;                 /* 0: RCCL collectives, 1: peer copies + ordered adds */
    uint64_t rays_primary, rays_bounce, samples;   /* over all GPUs */
    double seconds_render_max;         /* the slowest GPU's render time (host clock around its calls) */
    double seconds_reduce;             /* the exchange steps: accumulator reduce (+ min/max all-reduces in adaptive mode) */
    double seconds_total;
} hpt_group_stats;
int hpt_group_create(const int* devices, int n_devices, hpt_group** out);
void hpt_group_destroy(hpt_group* group);
int hpt_group_size(const hpt_group* group);
hpt_context* hpt_group_context(hpt_group* group, int rank);
int hpt_group_render(hpt_group* group, hpt_scene* scene, const hpt_render_params* params, float* rgb, uint32_t* samples_per_px,
                     hpt_group_stats* stats);

#ifdef __cplusplus
}
#endif
#endif
