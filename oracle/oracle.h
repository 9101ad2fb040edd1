/*
 * oracle.h -- CPU restatement of glowcoil/path-tracer's render path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (libhpt.so) never links, loads or calls it.
 *
 * PARITY UNPINNED: the reference is Rust, no cargo/rustc exists in the build image and the reference
 * ships no tests, golden vectors or output images.  This restatement follows
 * the reference sources line by line (citations at each function in oracle.c) and is pinned only by
 * analytic known-answer tests and structural facts of the shipped teapot (tests/test_oracle_*.py).
 * Third-party arithmetic restated from the crates' published behaviour (not in /root/reference):
 *   cgmath 0.15 (Matrix3::invert, normalize, dot order), wavefront_obj 5.1 (polygon fan, in the loader),
 *   rand 0.3 (replaced by the counter-based generator orc_rng_f32, spec shared with include/hpt.h).
 *
 * Conventions: all matrices are 3x3 COLUMN-major float[9] (m[c*3+r]) like cgmath::Matrix3;
 * all arithmetic is IEEE f32 compiled with -ffp-contract=off (rustc never fuses a*b+c).
 */
#ifndef ORACLE_H
#define ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_mesh orc_mesh;
typedef struct orc_scene orc_scene;

/* bvh.rs:13-23: pre-order node array. left/right = child indices (interior) or -1 (leaf); prim = leaf index or -1 */
typedef struct {
    int32_t left, right, prim;
    float p1[3], p2[3];
} orc_bvh_node;

enum { ORC_GEOM_NONE = 0, ORC_GEOM_SPHERE = 1, ORC_GEOM_PLANE = 2, ORC_GEOM_MESH = 3 };
enum { ORC_TEX_BLANK = 0, ORC_TEX_CHECKER = 1, ORC_TEX_IMAGE = 2 };

/* scene.rs:46-58 Texture + TextureData */
typedef struct {
    int32_t kind;          /* ORC_TEX_* */
    int32_t image;         /* image id for ORC_TEX_IMAGE */
    float color[3];
    float color1[3], color2[3];  /* checkerboard */
    float m[9], t[3];      /* texture transform (scene.rs:50) */
} orc_texture;

/* scene.rs:33-44 Material; only the fields the integrator reads (specular/glossiness/absorption are dead) */
typedef struct {
    orc_texture diffuse, reflection, refraction;
    float emission[3];
    float reflection_glossiness, refraction_glossiness, refraction_index;
} orc_material;

/* scene.rs:76-85 Camera as produced by load.rs:241-268 (up re-orthogonalised, NOT renormalised) */
typedef struct {
    float pos[3], dir[3], up[3];
    float fov;
    uint32_t img_width, img_height;
    float focaldist, dof;
} orc_camera;

typedef struct {
    int32_t adaptive;        /* 1: main.rs:67-107 (initial..max doubling); 0: fixed spp */
    int32_t spp;             /* fixed mode: samples per pixel */
    int32_t first_sample;    /* fixed mode: index of the first sample minus 1 (samples first_sample+1 ..= first_sample+spp) */
    int32_t initial_samples; /* adaptive: main.rs:31 (128) */
    int32_t max_samples;     /* adaptive: main.rs:32 (512) */
    float thresholds[3];     /* adaptive: main.rs:27-29 (.4,.3,.6) */
    int32_t max_depth;       /* deviation shared with the GPU core: bounces beyond this depth are cut */
    uint32_t seed;
    int32_t threads;         /* 0 = all */
} orc_render_params;

typedef struct {
    uint64_t rays;          /* Scene::intersect calls (primary + bounce segments) */
    uint64_t node_pops;     /* BVHIterator::next interior pops */
    uint64_t box_tests;     /* BoundingBox::intersect calls */
    uint64_t tri_tests;     /* intersect_triangle calls */
    double seconds;
    int32_t threads;
} orc_stats;

/* ---- sampling helpers ---- */
float orc_halton(int32_t index, int32_t base);                                   /* scene.rs:431 */
float orc_rng_f32(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t dim); /* counter-based; spec in include/hpt.h */
void orc_tonemap_rgba8(const float* rgb, uint32_t n_pixels, uint8_t* rgba);      /* scene.rs:362 */
int orc_mat3_invert(const float m[9], float out[9]);                             /* cgmath Matrix3::invert; 0 ok, -1 singular */

/* ---- bvh.rs ---- */
/* BVH::build (bvh.rs:35). boxes: n x 6 (p1,p2). Writes 2n-1 nodes. Returns node count or <0. */
int32_t orc_bvh_build(const float root_box[6], const float* boxes, int32_t n, orc_bvh_node* out_nodes);
/* BoundingBox::intersect (bvh.rs:225). Returns 1 and *t_in on hit. */
int orc_box_intersect(const float box[6], const float pos[3], const float dir[3], float* t_in);

/* ---- geometry.rs ---- */
/* Mesh as built by load.rs:282-321 (indices already 0-based). All index arrays are ntri x 3. uvw is nt x 3. */
orc_mesh* orc_mesh_create(const float* verts, int32_t nv, const uint32_t* tris,
                          const float* normals, int32_t nn, const uint32_t* ntris,
                          const float* uvw, int32_t nt, const uint32_t* ttris, int32_t ntri);
void orc_mesh_destroy(orc_mesh*);
int32_t orc_mesh_num_nodes(const orc_mesh*);
void orc_mesh_get_nodes(const orc_mesh*, orc_bvh_node* out);
void orc_mesh_get_box(const orc_mesh*, float box[6]);
/* Mesh::intersect (geometry.rs:132) for n local-space rays. prim=-1 on miss. side: 1 front, 0 back.
 * Note: like the reference this is the BVH candidate loop only (no Geometry::intersect mesh-box pre-test,
 * which is redundant with the root test at bvh.rs:149).  Optional outputs may be NULL. */
void orc_mesh_intersect(const orc_mesh*, int64_t n, const float* org3, const float* dir3,
                        int32_t* prim, float* t, float* u, float* v, uint8_t* side,
                        int32_t threads, orc_stats* stats);
/* intersect_triangle (geometry.rs:188). returns 1 on hit */
int orc_intersect_triangle(const float a[3], const float b[3], const float c[3], const float pos[3],
                           const float dir[3], float* t, float* u, float* v, int* side);

/* ---- scene.rs ---- */
orc_scene* orc_scene_create(void);
void orc_scene_destroy(orc_scene*);
/* mesh stays owned by the caller and must outlive the scene */
int32_t orc_scene_add_mesh(orc_scene*, const orc_mesh*);
int32_t orc_scene_add_image(orc_scene*, const uint8_t* rgb8, int32_t width, int32_t height);
int32_t orc_scene_add_material(orc_scene*, const orc_material*);
/* Node (scene.rs:19). parent=-1: top-level (Scene.nodes). geom ORC_GEOM_NONE = group without object. Returns node id. */
int32_t orc_scene_add_node(orc_scene*, int32_t parent, int32_t geom, int32_t mesh, int32_t material,
                           const float m[9], const float t[3]);
void orc_scene_set_background(orc_scene*, const orc_texture*);
void orc_scene_set_environment(orc_scene*, const orc_texture*);
void orc_scene_set_camera(orc_scene*, const orc_camera*);

/* Scene::intersect (scene.rs:190) for n world-space rays. node=-1 on miss. pos/normal/uv are the HitInfo
 * after all from_local_space steps. prim = triangle index for mesh hits else -1. Outputs may be NULL. */
void orc_scene_intersect(const orc_scene*, int64_t n, const float* org3, const float* dir3,
                         int32_t* node, int32_t* prim, float* t, uint8_t* side,
                         float* pos3, float* normal3, float* uvw3, int32_t threads, orc_stats* stats);

/* main.rs:43-115 minus the final tonemap: rgb = sum/num_samples per pixel (W*H*3 floats),
 * samples_per_px optional (W*H), per_sample optional (fixed mode only: W*H*spp*3, sample-major per pixel). */
int orc_render(const orc_scene*, const orc_render_params*, float* rgb, uint32_t* samples_per_px,
               float* per_sample, orc_stats* stats);
/* primary ray of (pixel x,y, sample i) exactly as main.rs:65-85: org3, dir3 */
void orc_primary_ray(const orc_scene*, uint32_t x, uint32_t y, int32_t i, uint32_t seed, float org[3], float dir[3]);

#ifdef __cplusplus
}
#endif
#endif
