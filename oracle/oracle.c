/*
 * oracle.c -- CPU restatement of glowcoil/path-tracer's render path.  TEST INFRASTRUCTURE ONLY
 * (see oracle.h: parity unpinned; never linked into or called from the product).
 *
 * Every function cites the reference lines it follows (paths relative to /root/reference).
 * Compile with -ffp-contract=off: the expressions below keep the reference's operation order.
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ small vector algebra (cgmath 0.15) */
typedef struct { float x, y, z; } v3;

static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
static inline v3 vscale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }   /* Vector3 * f32 and f32 * Vector3 */
static inline v3 vdivs(v3 a, float s) { return V(a.x / s, a.y / s, a.z / s); }    /* Vector3 / f32 */
static inline v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }  /* mul_element_wise */
/* cgmath dot = mul_element_wise().sum() = (x + y) + z */
static inline float vdot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline v3 vcross(v3 a, v3 b) {
    return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* cgmath normalize = self * (1 / magnitude) */
static inline v3 vnormalize(v3 a) { return vscale(a, 1.0f / sqrtf(vdot(a, a))); }

/* column-major 3x3: m[c*3+r] */
typedef struct { float m[9]; } m3;
#define M_(A, c, r) ((A).m[(c) * 3 + (r)])

/* Matrix3 * Vector3 = (row(0).dot(v), row(1).dot(v), row(2).dot(v)) */
static inline v3 m3mulv(const m3* a, v3 v) {
    return V((M_(*a, 0, 0) * v.x + M_(*a, 1, 0) * v.y) + M_(*a, 2, 0) * v.z,
             (M_(*a, 0, 1) * v.x + M_(*a, 1, 1) * v.y) + M_(*a, 2, 1) * v.z,
             (M_(*a, 0, 2) * v.x + M_(*a, 1, 2) * v.y) + M_(*a, 2, 2) * v.z);
}

/* cgmath Matrix3::invert: from_cols(c1 x c2 / det, c2 x c0 / det, c0 x c1 / det).transpose() */
static int m3invert(const m3* a, m3* out) {
    float det = M_(*a, 0, 0) * (M_(*a, 1, 1) * M_(*a, 2, 2) - M_(*a, 2, 1) * M_(*a, 1, 2)) -
                M_(*a, 1, 0) * (M_(*a, 0, 1) * M_(*a, 2, 2) - M_(*a, 2, 1) * M_(*a, 0, 2)) +
                M_(*a, 2, 0) * (M_(*a, 0, 1) * M_(*a, 1, 2) - M_(*a, 1, 1) * M_(*a, 0, 2));
    if (det == 0.0f) { for (int k = 0; k < 9; k++) out->m[k] = NAN; return -1; }  /* reference: unwrap() panics */
    v3 c0 = V(M_(*a, 0, 0), M_(*a, 0, 1), M_(*a, 0, 2));
    v3 c1 = V(M_(*a, 1, 0), M_(*a, 1, 1), M_(*a, 1, 2));
    v3 c2 = V(M_(*a, 2, 0), M_(*a, 2, 1), M_(*a, 2, 2));
    v3 r0 = vdivs(vcross(c1, c2), det);
    v3 r1 = vdivs(vcross(c2, c0), det);
    v3 r2 = vdivs(vcross(c0, c1), det);
    /* r0,r1,r2 are the columns before the transpose => rows of the result */
    M_(*out, 0, 0) = r0.x; M_(*out, 1, 0) = r0.y; M_(*out, 2, 0) = r0.z;
    M_(*out, 0, 1) = r1.x; M_(*out, 1, 1) = r1.y; M_(*out, 2, 1) = r1.z;
    M_(*out, 0, 2) = r2.x; M_(*out, 1, 2) = r2.y; M_(*out, 2, 2) = r2.z;
    return 0;
}
static inline m3 m3transpose(const m3* a) {
    m3 r;
    for (int c = 0; c < 3; c++) for (int k = 0; k < 3; k++) M_(r, c, k) = M_(*a, k, c);
    return r;
}
int orc_mat3_invert(const float m[9], float out[9]) {
    m3 a, b; memcpy(a.m, m, sizeof a.m);
    if (m3invert(&a, &b)) return -1;
    memcpy(out, b.m, sizeof b.m);
    return 0;
}

/* ------------------------------------------------------------------ counters */
static _Thread_local orc_stats* tl_stats = NULL;
#define COUNT(field) do { if (tl_stats) tl_stats->field++; } while (0)

static double now_s(void) {
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
static int pick_threads(int req) {
#ifdef _OPENMP
    return req > 0 ? req : omp_get_max_threads();
#else
    (void)req; return 1;
#endif
}
static void stats_merge(orc_stats* dst, const orc_stats* src) {
    dst->rays += src->rays; dst->node_pops += src->node_pops;
    dst->box_tests += src->box_tests; dst->tri_tests += src->tri_tests;
}

/* ------------------------------------------------------------------ sampling helpers */
/* scene.rs:431-443 */
float orc_halton(int32_t index, int32_t base) {
    float r = 0.0f, f = 1.0f;
    int32_t i = index;
    while (i > 0) {
        f /= (float)base;
        r += f * (float)(i % base);
        i /= base;
    }
    return r;
}

/* Counter-based generator replacing rand::random::<f32>() (rand 0.3: 24-bit uniform in [0,1)).
 * Spec (shared with include/hpt.h, restated here on purpose so the checker stays independent):
 *   mix = lowbias32; k1 = mix(pixel + mix(seed)); k2 = mix(sample ^ 0x5bd1e995 ^ k-seed)
 *   u   = mix(k1 ^ (k2 + dim * 0x9E3779B9)) >> 8, scaled by 2^-24 */
static inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
float orc_rng_f32(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t dim) {
    uint32_t k1 = mix32(pixel + mix32(seed));
    uint32_t k2 = mix32(sample ^ 0x5bd1e995U ^ (seed * 0x9E3779B9U));
    uint32_t h = mix32(k1 ^ (k2 + dim * 0x9E3779B9U));
    return (float)(h >> 8) * (1.0f / 16777216.0f);
}

/* scene.rs:360-367. Rust's f32::max/min return the non-NaN operand; `as u8` truncates (saturating). */
static inline float rmaxf(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a > b ? a : b)); }
static inline float rminf(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a < b ? a : b)); }
void orc_tonemap_rgba8(const float* rgb, uint32_t n, uint8_t* rgba) {
    const float GAMMA = 1.0f / 2.2f;
    for (uint32_t i = 0; i < n; i++) {
        for (int c = 0; c < 3; c++) {
            float v = rminf(rmaxf(powf(rgb[3 * i + c], GAMMA) * 255.0f, 0.0f), 255.0f);
            rgba[4 * i + c] = (uint8_t)v;
        }
        rgba[4 * i + 3] = 255;
    }
}

/* ------------------------------------------------------------------ bvh.rs */
typedef struct { v3 p1, p2; } bbox;

/* bvh.rs:225-281 */
static int box_intersect(const bbox* b, v3 pos, v3 dir, float* t_in_out) {
    COUNT(box_tests);
    float in_x = -INFINITY, out_x = INFINITY;
    if (dir.x == 0.0f) {
        if (pos.x < b->p1.x || pos.x > b->p2.x) return 0;
    } else {
        in_x = (b->p1.x - pos.x) / dir.x;
        out_x = (b->p2.x - pos.x) / dir.x;
        if (out_x < in_x) { float tmp = in_x; in_x = out_x; out_x = tmp; }
    }
    float in_y = -INFINITY, out_y = INFINITY;
    if (dir.y == 0.0f) {
        if (pos.y < b->p1.y || pos.y > b->p2.y) return 0;
    } else {
        in_y = (b->p1.y - pos.y) / dir.y;
        out_y = (b->p2.y - pos.y) / dir.y;
        if (out_y < in_y) { float tmp = in_y; in_y = out_y; out_y = tmp; }
    }
    float in_z = -INFINITY, out_z = INFINITY;
    if (dir.z == 0.0f) {
        if (pos.z < b->p1.z || pos.z > b->p2.z) return 0;
    } else {
        in_z = (b->p1.z - pos.z) / dir.z;
        out_z = (b->p2.z - pos.z) / dir.z;
        if (out_z < in_z) { float tmp = in_z; in_z = out_z; out_z = tmp; }
    }
    float t_in = rmaxf(rmaxf(in_x, in_y), in_z);
    float t_out = rminf(rminf(out_x, out_y), out_z);
    if (t_in <= t_out) { *t_in_out = t_in; return 1; }
    return 0;
}
int orc_box_intersect(const float box[6], const float pos[3], const float dir[3], float* t_in) {
    bbox b = { V(box[0], box[1], box[2]), V(box[3], box[4], box[5]) };
    float t = 0.0f;
    int r = box_intersect(&b, V(pos[0], pos[1], pos[2]), V(dir[0], dir[1], dir[2]), &t);
    if (r && t_in) *t_in = t;
    return r;
}
/* bvh.rs:283-291 (f32::min/max) */
static void box_union(bbox* a, const bbox* o) {
    a->p1.x = rminf(a->p1.x, o->p1.x); a->p1.y = rminf(a->p1.y, o->p1.y); a->p1.z = rminf(a->p1.z, o->p1.z);
    a->p2.x = rmaxf(a->p2.x, o->p2.x); a->p2.y = rmaxf(a->p2.y, o->p2.y); a->p2.z = rmaxf(a->p2.z, o->p2.z);
}

typedef struct { int32_t left, right, prim; bbox box; } bnode;
typedef struct { bnode* nodes; int32_t n, cap; } bvh_t;

static int32_t bvh_push(bvh_t* b, bnode nd) {
    if (b->n == b->cap) { b->cap = b->cap ? b->cap * 2 : 64; b->nodes = (bnode*)realloc(b->nodes, (size_t)b->cap * sizeof(bnode)); }
    b->nodes[b->n] = nd;
    return b->n++;
}

/* bvh.rs:48-146 (both quirks kept: half-extent "center", parent box handed to the right child) */
static void bvh_split(bvh_t* b, int32_t* elems, int32_t len, const bbox* root_box, const bbox* boxes) {
    if (len == 1) {
        bnode leaf = { -1, -1, elems[0], boxes[elems[0]] };
        bvh_push(b, leaf);
        return;
    }
    float x_size = root_box->p2.x - root_box->p1.x;
    float y_size = root_box->p2.y - root_box->p1.y;
    float z_size = root_box->p2.z - root_box->p1.z;
    int32_t j = 0;
    if (x_size > y_size && x_size > z_size) {
        float pivot = root_box->p1.x + x_size / 2.0f;
        for (int32_t i = 0; i < len; i++) {
            const bbox* eb = &boxes[elems[i]];
            float center = (eb->p2.x - eb->p1.x) / 2.0f;
            if (center < pivot) { int32_t tmp = elems[i]; elems[i] = elems[j]; elems[j] = tmp; j++; }
        }
    } else if (y_size > z_size) {
        float pivot = root_box->p1.y + y_size / 2.0f;
        for (int32_t i = 0; i < len; i++) {
            const bbox* eb = &boxes[elems[i]];
            float center = (eb->p2.y - eb->p1.y) / 2.0f;
            if (center < pivot) { int32_t tmp = elems[i]; elems[i] = elems[j]; elems[j] = tmp; j++; }
        }
    } else {
        float pivot = root_box->p1.z + z_size / 2.0f;
        for (int32_t i = 0; i < len; i++) {
            const bbox* eb = &boxes[elems[i]];
            float center = (eb->p2.z - eb->p1.z) / 2.0f;
            if (center < pivot) { int32_t tmp = elems[i]; elems[i] = elems[j]; elems[j] = tmp; j++; }
        }
    }
    if (j == 0 || j == len) j = len / 2;

    bnode nd = { 0, 0, -1, *root_box };
    int32_t node_index = bvh_push(b, nd);
    b->nodes[node_index].left = node_index + 1;
    {
        bbox left_box = boxes[elems[0]];
        for (int32_t k = 1; k < j; k++) box_union(&left_box, &boxes[elems[k]]);
        bvh_split(b, elems, j, &left_box, boxes);
    }
    b->nodes[node_index].right = b->n;
    {
        /* bvh.rs:139-144: the tight right box is computed and then NOT used */
        bvh_split(b, elems + j, len - j, root_box, boxes);
    }
}

static int bvh_build(bvh_t* b, const bbox* root_box, const bbox* boxes, int32_t n) {
    b->nodes = NULL; b->n = 0; b->cap = 0;
    if (n <= 0) return -1;  /* bvh.rs:36-38 panics */
    int32_t* elems = (int32_t*)malloc((size_t)n * sizeof(int32_t));
    for (int32_t i = 0; i < n; i++) elems[i] = i;
    bvh_split(b, elems, n, root_box, boxes);
    free(elems);
    return 0;
}

int32_t orc_bvh_build(const float root_box[6], const float* boxes, int32_t n, orc_bvh_node* out) {
    if (n <= 0) return -1;
    bbox rb = { V(root_box[0], root_box[1], root_box[2]), V(root_box[3], root_box[4], root_box[5]) };
    bbox* bx = (bbox*)malloc((size_t)(n > 0 ? n : 1) * sizeof(bbox));
    for (int32_t i = 0; i < n; i++) {
        bx[i].p1 = V(boxes[6 * i], boxes[6 * i + 1], boxes[6 * i + 2]);
        bx[i].p2 = V(boxes[6 * i + 3], boxes[6 * i + 4], boxes[6 * i + 5]);
    }
    bvh_t b;
    if (bvh_build(&b, &rb, bx, n)) { free(bx); return -1; }
    for (int32_t i = 0; i < b.n; i++) {
        out[i].left = b.nodes[i].left; out[i].right = b.nodes[i].right; out[i].prim = b.nodes[i].prim;
        out[i].p1[0] = b.nodes[i].box.p1.x; out[i].p1[1] = b.nodes[i].box.p1.y; out[i].p1[2] = b.nodes[i].box.p1.z;
        out[i].p2[0] = b.nodes[i].box.p2.x; out[i].p2[1] = b.nodes[i].box.p2.y; out[i].p2[2] = b.nodes[i].box.p2.z;
    }
    int32_t n_nodes = b.n;
    free(b.nodes); free(bx);
    return n_nodes;
}

/* ------------------------------------------------------------------ geometry.rs */
struct orc_mesh {
    v3* vertices; int32_t nv;
    v3* normals; int32_t nn;
    v3* tex; int32_t nt;
    uint32_t *tris, *ntris, *ttris; int32_t ntri;
    bbox box;
    bvh_t bvh;
};

typedef struct { float z; v3 pos, uv, normal; int side; int32_t prim; float bu, bv; } hitinfo; /* scene.rs:87-93 (+prim, barycentrics for tests) */

/* geometry.rs:188-221 */
static int intersect_triangle(v3 a, v3 b, v3 c, v3 pos, v3 dir, float* t_, float* u_, float* v_, int* side_) {
    COUNT(tri_tests);
    const float EPSILON = 1.0e-8f;  /* scene.rs:123 */
    v3 ab = vsub(b, a);
    v3 ac = vsub(c, a);
    v3 pvec = vcross(dir, ac);
    float det = vdot(ab, pvec);
    if (fabsf(det) < EPSILON) return 0;
    v3 tvec = vsub(pos, a);
    float u = vdot(tvec, pvec) / det;
    if (u < 0.0f || u > 1.0f) return 0;
    v3 qvec = vcross(tvec, ab);
    float v = vdot(dir, qvec) / det;
    if (v < 0.0f || u + v > 1.0f) return 0;
    float t = vdot(ac, qvec) / det;
    if (t < 0.0f) return 0;
    *t_ = t; *u_ = u; *v_ = v; *side_ = det > 0.0f ? 1 : 0;
    return 1;
}
int orc_intersect_triangle(const float a[3], const float b[3], const float c[3], const float pos[3],
                           const float dir[3], float* t, float* u, float* v, int* side) {
    float tt, uu, vv; int ss;
    int r = intersect_triangle(V(a[0], a[1], a[2]), V(b[0], b[1], b[2]), V(c[0], c[1], c[2]),
                               V(pos[0], pos[1], pos[2]), V(dir[0], dir[1], dir[2]), &tt, &uu, &vv, &ss);
    if (r) { *t = tt; *u = uu; *v = vv; *side = ss; }
    return r;
}

/* geometry.rs:117-130: (1-u-v)*A + u*B + v*C, left to right */
static inline v3 bary(const v3* arr, const uint32_t* idx, float u, float v) {
    float w = 1.0f - u - v;
    return vadd(vadd(vscale(arr[idx[0]], w), vscale(arr[idx[1]], u)), vscale(arr[idx[2]], v));
}

/* geometry.rs:132-158 with BVH::traverse / BVHIterator::next (bvh.rs:148-204) inlined */
static int mesh_intersect(const orc_mesh* m, v3 pos, v3 dir, hitinfo* out) {
    const bnode* nodes = m->bvh.nodes;
    int have = 0;
    hitinfo nearest; memset(&nearest, 0, sizeof nearest);
    int32_t stack_local[128];
    int32_t* stack = stack_local; int32_t cap = 128, sp = 0;
    float tmp;
    if (box_intersect(&nodes[0].box, pos, dir, &tmp)) stack[sp++] = 0;   /* bvh.rs:149-153 */
    while (sp > 0) {
        int32_t i = stack[--sp];
        const bnode* nd = &nodes[i];
        if (nd->prim < 0) {
            COUNT(node_pops);
            float t_left = 0.0f, t_right = 0.0f;
            int left = box_intersect(&nodes[nd->left].box, pos, dir, &t_left);
            int right = box_intersect(&nodes[nd->right].box, pos, dir, &t_right);
            if (sp + 2 > cap) {
                int32_t* ns = (int32_t*)malloc((size_t)cap * 2 * sizeof(int32_t));
                memcpy(ns, stack, (size_t)sp * sizeof(int32_t));
                if (stack != stack_local) free(stack);
                stack = ns; cap *= 2;
            }
            if (left) {
                if (right) {
                    int32_t first, second;
                    if (t_left < t_right) { first = nd->left; second = nd->right; }
                    else { first = nd->right; second = nd->left; }
                    stack[sp++] = second;
                    stack[sp++] = first;
                } else stack[sp++] = nd->left;
            } else if (right) stack[sp++] = nd->right;
        } else {
            int32_t f = nd->prim;
            const uint32_t* tri = &m->tris[3 * f];
            float t, u, v; int side;
            if (intersect_triangle(m->vertices[tri[0]], m->vertices[tri[1]], m->vertices[tri[2]], pos, dir, &t, &u, &v, &side)) {
                if (have && nearest.z < t) continue;   /* geometry.rs:142-146: later candidate wins ties */
                have = 1;
                nearest.z = t;
                nearest.pos = bary(m->vertices, tri, u, v);
                nearest.uv = bary(m->tex, &m->ttris[3 * f], u, v);
                nearest.normal = bary(m->normals, &m->ntris[3 * f], u, v);
                nearest.side = side;
                nearest.prim = f;
                nearest.bu = u; nearest.bv = v;
            }
        }
    }
    if (stack != stack_local) free(stack);
    if (have) *out = nearest;
    return have;
}

orc_mesh* orc_mesh_create(const float* verts, int32_t nv, const uint32_t* tris,
                          const float* normals, int32_t nn, const uint32_t* ntris,
                          const float* uvw, int32_t nt, const uint32_t* ttris, int32_t ntri) {
    if (nv <= 0 || ntri <= 0) return NULL;
    orc_mesh* m = (orc_mesh*)calloc(1, sizeof *m);
    m->nv = nv; m->nn = nn; m->nt = nt; m->ntri = ntri;
    m->vertices = (v3*)malloc((size_t)nv * sizeof(v3));
    m->normals = (v3*)malloc((size_t)(nn > 0 ? nn : 1) * sizeof(v3));
    m->tex = (v3*)malloc((size_t)(nt > 0 ? nt : 1) * sizeof(v3));
    memcpy(m->vertices, verts, (size_t)nv * sizeof(v3));
    if (nn > 0) memcpy(m->normals, normals, (size_t)nn * sizeof(v3));
    if (nt > 0) memcpy(m->tex, uvw, (size_t)nt * sizeof(v3));
    size_t ib = (size_t)ntri * 3 * sizeof(uint32_t);
    m->tris = (uint32_t*)malloc(ib); m->ntris = (uint32_t*)malloc(ib); m->ttris = (uint32_t*)malloc(ib);
    memcpy(m->tris, tris, ib); memcpy(m->ntris, ntris, ib); memcpy(m->ttris, ttris, ib);
    /* load.rs:284-293: strict < / > against the running box */
    v3 p1 = m->vertices[0], p2 = m->vertices[0];
    for (int32_t i = 0; i < nv; i++) {
        v3 q = m->vertices[i];
        if (q.x < p1.x) p1.x = q.x;
        if (q.y < p1.y) p1.y = q.y;
        if (q.z < p1.z) p1.z = q.z;
        if (q.x > p2.x) p2.x = q.x;
        if (q.y > p2.y) p2.y = q.y;
        if (q.z > p2.z) p2.z = q.z;
    }
    m->box.p1 = p1; m->box.p2 = p2;
    /* geometry.rs:223-239 */
    bbox* boxes = (bbox*)malloc((size_t)ntri * sizeof(bbox));
    for (int32_t f = 0; f < ntri; f++) {
        v3 a = m->vertices[tris[3 * f]], b = m->vertices[tris[3 * f + 1]], c = m->vertices[tris[3 * f + 2]];
        bbox bb = { a, a }, bb2 = { b, b }, bb3 = { c, c };
        box_union(&bb, &bb2); box_union(&bb, &bb3);
        boxes[f] = bb;
    }
    bvh_build(&m->bvh, &m->box, boxes, ntri);
    free(boxes);
    return m;
}
void orc_mesh_destroy(orc_mesh* m) {
    if (!m) return;
    free(m->vertices); free(m->normals); free(m->tex); free(m->tris); free(m->ntris); free(m->ttris);
    free(m->bvh.nodes); free(m);
}
int32_t orc_mesh_num_nodes(const orc_mesh* m) { return m->bvh.n; }
void orc_mesh_get_nodes(const orc_mesh* m, orc_bvh_node* out) {
    for (int32_t i = 0; i < m->bvh.n; i++) {
        const bnode* b = &m->bvh.nodes[i];
        out[i].left = b->left; out[i].right = b->right; out[i].prim = b->prim;
        out[i].p1[0] = b->box.p1.x; out[i].p1[1] = b->box.p1.y; out[i].p1[2] = b->box.p1.z;
        out[i].p2[0] = b->box.p2.x; out[i].p2[1] = b->box.p2.y; out[i].p2[2] = b->box.p2.z;
    }
}
void orc_mesh_get_box(const orc_mesh* m, float box[6]) {
    box[0] = m->box.p1.x; box[1] = m->box.p1.y; box[2] = m->box.p1.z;
    box[3] = m->box.p2.x; box[4] = m->box.p2.y; box[5] = m->box.p2.z;
}

void orc_mesh_intersect(const orc_mesh* m, int64_t n, const float* org3, const float* dir3,
                        int32_t* prim, float* t, float* u, float* v, uint8_t* side,
                        int32_t threads, orc_stats* stats) {
    int nt = pick_threads(threads);
    double t0 = now_s();
    orc_stats total; memset(&total, 0, sizeof total);
#pragma omp parallel num_threads(nt)
    {
        orc_stats local; memset(&local, 0, sizeof local);
        tl_stats = stats ? &local : NULL;
#pragma omp for schedule(dynamic, 4096)
        for (int64_t i = 0; i < n; i++) {
            hitinfo h;
            local.rays++;
            int r = mesh_intersect(m, V(org3[3 * i], org3[3 * i + 1], org3[3 * i + 2]),
                                   V(dir3[3 * i], dir3[3 * i + 1], dir3[3 * i + 2]), &h);
            if (prim) prim[i] = r ? h.prim : -1;
            if (t) t[i] = r ? h.z : 0.0f;
            if (u) u[i] = r ? h.bu : 0.0f;
            if (v) v[i] = r ? h.bv : 0.0f;
            if (side) side[i] = r ? (uint8_t)h.side : 0;
        }
        tl_stats = NULL;
#pragma omp critical
        stats_merge(&total, &local);
    }
    if (stats) { *stats = total; stats->seconds = now_s() - t0; stats->threads = nt; }
}

/* ------------------------------------------------------------------ scene.rs */
typedef struct { m3 m; v3 t; } xform;  /* scene.rs:102-105 */

typedef struct {
    int32_t geom, mesh, material, parent;
    xform xf;
    int32_t* children; int32_t nchildren, capchildren;
} onode;

typedef struct { uint8_t* px; int32_t w, h; } oimage;

struct orc_scene {
    onode* nodes; int32_t nnodes, capnodes;
    int32_t* top; int32_t ntop, captop;
    const orc_mesh** meshes; int32_t nmeshes;
    oimage* images; int32_t nimages;
    orc_material* materials; int32_t nmaterials;
    orc_texture background, environment;
    orc_camera camera;
};

static void tex_default(orc_texture* t) {   /* load.rs:30-34: Blank, colour 0, identity transform */
    memset(t, 0, sizeof *t);
    t->m[0] = t->m[4] = t->m[8] = 1.0f;
}
orc_scene* orc_scene_create(void) {
    orc_scene* s = (orc_scene*)calloc(1, sizeof *s);
    tex_default(&s->background); tex_default(&s->environment);
    /* scene.rs:107-120 */
    orc_camera c = { {0, 0, 0}, {0, 0, -1}, {0, 1, 0}, 40.0f, 800, 600, 1.0f, 0.0f };
    s->camera = c;
    return s;
}
void orc_scene_destroy(orc_scene* s) {
    if (!s) return;
    for (int32_t i = 0; i < s->nnodes; i++) free(s->nodes[i].children);
    for (int32_t i = 0; i < s->nimages; i++) free(s->images[i].px);
    free(s->nodes); free(s->top); free(s->meshes); free(s->images); free(s->materials); free(s);
}
int32_t orc_scene_add_mesh(orc_scene* s, const orc_mesh* m) {
    s->meshes = (const orc_mesh**)realloc(s->meshes, (size_t)(s->nmeshes + 1) * sizeof(*s->meshes));
    s->meshes[s->nmeshes] = m;
    return s->nmeshes++;
}
int32_t orc_scene_add_image(orc_scene* s, const uint8_t* rgb8, int32_t w, int32_t h) {
    s->images = (oimage*)realloc(s->images, (size_t)(s->nimages + 1) * sizeof(oimage));
    oimage* im = &s->images[s->nimages];
    size_t bytes = (size_t)w * (size_t)h * 3;
    im->px = (uint8_t*)malloc(bytes ? bytes : 1);
    if (bytes) memcpy(im->px, rgb8, bytes);
    im->w = w; im->h = h;
    return s->nimages++;
}
int32_t orc_scene_add_material(orc_scene* s, const orc_material* m) {
    s->materials = (orc_material*)realloc(s->materials, (size_t)(s->nmaterials + 1) * sizeof(orc_material));
    s->materials[s->nmaterials] = *m;
    return s->nmaterials++;
}
static void push_i32(int32_t** arr, int32_t* n, int32_t* cap, int32_t v) {
    if (*n == *cap) { *cap = *cap ? *cap * 2 : 8; *arr = (int32_t*)realloc(*arr, (size_t)*cap * sizeof(int32_t)); }
    (*arr)[(*n)++] = v;
}
int32_t orc_scene_add_node(orc_scene* s, int32_t parent, int32_t geom, int32_t mesh, int32_t material,
                           const float m[9], const float t[3]) {
    if (s->nnodes == s->capnodes) {
        s->capnodes = s->capnodes ? s->capnodes * 2 : 16;
        s->nodes = (onode*)realloc(s->nodes, (size_t)s->capnodes * sizeof(onode));
    }
    onode* nd = &s->nodes[s->nnodes];
    memset(nd, 0, sizeof *nd);
    nd->geom = geom; nd->mesh = mesh; nd->material = material; nd->parent = parent;
    memcpy(nd->xf.m.m, m, sizeof nd->xf.m.m);
    nd->xf.t = V(t[0], t[1], t[2]);
    int32_t id = s->nnodes++;
    if (parent < 0) push_i32(&s->top, &s->ntop, &s->captop, id);
    else push_i32(&s->nodes[parent].children, &s->nodes[parent].nchildren, &s->nodes[parent].capchildren, id);
    return id;
}
void orc_scene_set_background(orc_scene* s, const orc_texture* t) { s->background = *t; }
void orc_scene_set_environment(orc_scene* s, const orc_texture* t) { s->environment = *t; }
void orc_scene_set_camera(orc_scene* s, const orc_camera* c) { s->camera = *c; }

/* scene.rs:278-288: the inverse is recomputed on every call, exactly like the reference */
static inline v3 xf_to_local(const xform* x, v3 v) {
    m3 inv; m3invert(&x->m, &inv);
    return m3mulv(&inv, vsub(v, x->t));
}
static inline v3 xf_from_local(const xform* x, v3 v) { return vadd(m3mulv(&x->m, v), x->t); }
static inline v3 xf_normal_from_local(const xform* x, v3 n) {
    m3 inv; m3invert(&x->m, &inv);
    m3 it = m3transpose(&inv);
    return vnormalize(m3mulv(&it, n));
}

#define PI_F 3.14159265358979323846f

/* geometry.rs:30-97 */
static int geometry_intersect(const orc_scene* s, const onode* nd, v3 pos, v3 dir, hitinfo* out) {
    bbox bb;
    switch (nd->geom) {   /* geometry.rs:100-114 */
        case ORC_GEOM_SPHERE: bb.p1 = V(-1, -1, -1); bb.p2 = V(1, 1, 1); break;
        case ORC_GEOM_PLANE: bb.p1 = V(-1, -1, 0); bb.p2 = V(1, 1, 0); break;
        default: bb = s->meshes[nd->mesh]->box; break;
    }
    float tmp;
    if (!box_intersect(&bb, pos, dir, &tmp)) return 0;   /* geometry.rs:31 */

    if (nd->geom == ORC_GEOM_SPHERE) {   /* geometry.rs:36-73 */
        float pos_dot_dir = vdot(pos, dir);
        float dir_len_sqr = vdot(dir, dir);
        float discriminant = 4.0f * (pos_dot_dir * pos_dot_dir) - 4.0f * dir_len_sqr * (vdot(pos, pos) - 1.0f);
        if (discriminant > 0.0f) {
            float ds = sqrtf(discriminant);
            float t1 = (-2.0f * pos_dot_dir + ds) / (2.0f * dir_len_sqr);
            float t2 = (-2.0f * pos_dot_dir - ds) / (2.0f * dir_len_sqr);
            if (t1 > 0.0f || t2 > 0.0f) {
                float t; int side;
                if (t1 > 0.0f && t2 > 0.0f) { side = 1; t = rminf(t1, t2); }
                else { side = 0; t = rmaxf(t1, t2); }
                v3 hit_pos = vadd(pos, vscale(dir, t));
                v3 normal = vnormalize(hit_pos);
                out->z = t; out->pos = hit_pos;
                out->uv = V(0.5f + atan2f(normal.y, normal.x) / (2.0f * PI_F), 0.5f - asinf(-normal.z) / PI_F, 0.0f);
                out->normal = normal; out->side = side; out->prim = -1;
                return 1;
            }
        }
        return 0;
    } else if (nd->geom == ORC_GEOM_PLANE) {   /* geometry.rs:74-92 */
        float t = -(pos.z / dir.z);
        if (t > 0.0f) {
            v3 p = vadd(pos, vscale(dir, t));
            if (-1.0f < p.x && p.x < 1.0f && -1.0f < p.y && p.y < 1.0f) {
                out->z = t; out->pos = p;
                out->uv = V(0.5f + 0.5f * p.x, 0.5f + 0.5f * p.y, 0.0f);
                out->normal = V(0, 0, 1);
                out->side = pos.z > 0.0f ? 1 : 0; out->prim = -1;
                return 1;
            }
        }
        return 0;
    }
    return mesh_intersect(s->meshes[nd->mesh], pos, dir, out);   /* geometry.rs:93-95 */
}

/* scene.rs:242-275 */
static int node_intersect(const orc_scene* s, int32_t id, v3 pos, v3 dir, hitinfo* out, int32_t* out_node) {
    const onode* nd = &s->nodes[id];
    /* scene.rs:228-232 */
    v3 local_pos = xf_to_local(&nd->xf, pos);
    v3 local_dir = vsub(xf_to_local(&nd->xf, vadd(pos, dir)), local_pos);

    int have = 0; hitinfo nearest; int32_t nearest_node = -1;
    memset(&nearest, 0, sizeof nearest);
    if (nd->geom != ORC_GEOM_NONE) {
        if (geometry_intersect(s, nd, local_pos, local_dir, &nearest)) { have = 1; nearest_node = id; }
    }
    for (int32_t k = 0; k < nd->nchildren; k++) {
        hitinfo h; int32_t hn;
        if (node_intersect(s, nd->children[k], local_pos, local_dir, &h, &hn)) {
            if (have) {
                if (h.z < nearest.z) { nearest = h; nearest_node = hn; }   /* scene.rs:251: earlier wins ties */
            } else { have = 1; nearest = h; nearest_node = hn; }
        }
    }
    if (!have) return 0;
    out->z = nearest.z;
    out->pos = xf_from_local(&nd->xf, nearest.pos);
    out->uv = nearest.uv;
    out->normal = xf_normal_from_local(&nd->xf, nearest.normal);
    out->side = nearest.side;
    out->prim = nearest.prim;
    *out_node = nearest_node;
    return 1;
}

/* scene.rs:190-208 */
static int scene_intersect(const orc_scene* s, v3 pos, v3 dir, hitinfo* out, int32_t* out_node) {
    COUNT(rays);
    int have = 0;
    for (int32_t k = 0; k < s->ntop; k++) {
        hitinfo h; int32_t hn;
        if (node_intersect(s, s->top[k], pos, dir, &h, &hn)) {
            if (have) {
                if (h.z < out->z) { *out = h; *out_node = hn; }
            } else { have = 1; *out = h; *out_node = hn; }
        }
    }
    return have;
}

void orc_scene_intersect(const orc_scene* s, int64_t n, const float* org3, const float* dir3,
                         int32_t* node, int32_t* prim, float* t, uint8_t* side,
                         float* pos3, float* normal3, float* uvw3, int32_t threads, orc_stats* stats) {
    int nt = pick_threads(threads);
    double t0 = now_s();
    orc_stats total; memset(&total, 0, sizeof total);
#pragma omp parallel num_threads(nt)
    {
        orc_stats local; memset(&local, 0, sizeof local);
        tl_stats = &local;
#pragma omp for schedule(dynamic, 1024)
        for (int64_t i = 0; i < n; i++) {
            hitinfo h; int32_t hn = -1;
            memset(&h, 0, sizeof h);
            int r = scene_intersect(s, V(org3[3 * i], org3[3 * i + 1], org3[3 * i + 2]),
                                    V(dir3[3 * i], dir3[3 * i + 1], dir3[3 * i + 2]), &h, &hn);
            if (node) node[i] = r ? hn : -1;
            if (prim) prim[i] = r ? h.prim : -1;
            if (t) t[i] = r ? h.z : 0.0f;
            if (side) side[i] = r ? (uint8_t)h.side : 0;
            if (pos3) { pos3[3 * i] = r ? h.pos.x : 0; pos3[3 * i + 1] = r ? h.pos.y : 0; pos3[3 * i + 2] = r ? h.pos.z : 0; }
            if (normal3) { normal3[3 * i] = r ? h.normal.x : 0; normal3[3 * i + 1] = r ? h.normal.y : 0; normal3[3 * i + 2] = r ? h.normal.z : 0; }
            if (uvw3) { uvw3[3 * i] = r ? h.uv.x : 0; uvw3[3 * i + 1] = r ? h.uv.y : 0; uvw3[3 * i + 2] = r ? h.uv.z : 0; }
        }
        tl_stats = NULL;
#pragma omp critical
        stats_merge(&total, &local);
    }
    if (stats) { *stats = total; stats->seconds = now_s() - t0; stats->threads = nt; }
}

/* scene.rs:378-392 (`as i32` truncates toward zero, saturating; NaN -> 0) */
static inline float trunc_i32(float x) {
    if (x != x) return 0.0f;
    if (x >= 2147483648.0f) return 2147483648.0f;   /* i32::MAX as f32 rounds to 2^31 */
    if (x <= -2147483648.0f) return -2147483648.0f;
    return (float)(int32_t)x;
}
static v3 unit_clamp(v3 p) {
    float x = p.x - trunc_i32(p.x); if (x < 0.0f) x += 1.0f;
    float y = p.y - trunc_i32(p.y); if (y < 0.0f) y += 1.0f;
    float z = p.z - trunc_i32(p.z); if (z < 0.0f) z += 1.0f;
    return V(x, y, z);
}
/* scene.rs:369-376 */
static inline v3 u8_as_color(const uint8_t* c) {
    return V(rminf(rmaxf((float)c[0] / 255.0f, 0.0f), 1.0f), rminf(rmaxf((float)c[1] / 255.0f, 0.0f), 1.0f),
             rminf(rmaxf((float)c[2] / 255.0f, 0.0f), 1.0f));
}
/* scene.rs:312-358 */
static v3 texdata_sample(const orc_scene* s, const orc_texture* tx, v3 point) {
    if (tx->kind == ORC_TEX_BLANK) return V(1, 1, 1);
    if (tx->kind == ORC_TEX_IMAGE) {
        const oimage* im = &s->images[tx->image];
        size_t width = (size_t)im->w, height = (size_t)im->h;
        if (width + height == 0) return V(0, 0, 0);
        v3 c = unit_clamp(point);
        float x = c.x * (float)width;
        float y = c.y * (float)height;
        size_t x1 = (size_t)x, y1 = (size_t)y;   /* x,y >= 0 here */
        float xt = x - (float)x1, yt = y - (float)y1;
        x1 = x1 % width; y1 = y1 % height;
        size_t x2 = (x1 + 1) % width, y2 = (y1 + 1) % height;
        v3 p00 = u8_as_color(&im->px[3 * (y1 * width + x1)]);
        v3 p10 = u8_as_color(&im->px[3 * (y1 * width + x2)]);
        v3 p01 = u8_as_color(&im->px[3 * (y2 * width + x1)]);
        v3 p11 = u8_as_color(&im->px[3 * (y2 * width + x2)]);
        /* scene.rs:346: ((a*p00 + b*p10) + c*p01) + d*p11 */
        v3 r = vscale(p00, (1.0f - xt) * (1.0f - yt));
        r = vadd(r, vscale(p10, xt * (1.0f - yt)));
        r = vadd(r, vscale(p01, (1.0f - xt) * yt));
        r = vadd(r, vscale(p11, xt * yt));
        return r;
    }
    v3 c = unit_clamp(point);   /* scene.rs:348-355 */
    if ((c.x < 0.5f && c.y < 0.5f) || (c.x > 0.5f && c.y > 0.5f)) return V(tx->color1[0], tx->color1[1], tx->color1[2]);
    return V(tx->color2[0], tx->color2[1], tx->color2[2]);
}
/* scene.rs:299-301 */
static v3 texture_sample(const orc_scene* s, const orc_texture* tx, v3 point) {
    xform x; memcpy(x.m.m, tx->m, sizeof x.m.m); x.t = V(tx->t[0], tx->t[1], tx->t[2]);
    return vmul(V(tx->color[0], tx->color[1], tx->color[2]), texdata_sample(s, tx, xf_to_local(&x, point)));
}
/* scene.rs:303-305 */
static v3 texture_sample_environment(const orc_scene* s, const orc_texture* tx, v3 dir) {
    return texture_sample(s, tx, V(0.5f + atan2f(dir.y, dir.x) / (2.0f * PI_F), 0.5f - asinf(-dir.z) / PI_F, 0.0f));
}

/* scene.rs:211-213 */
static v3 reflect_ray(v3 vec, v3 normal) {
    return vnormalize(vadd(vneg(vec), vscale(normal, 2.0f * vdot(normal, vec))));
}
/* scene.rs:215-225 */
static int refract_ray(v3 vec, v3 normal, float n1, float n2, v3* out) {
    float n = n1 / n2;
    float ndv = vdot(normal, vec);
    v3 s = vscale(vsub(vscale(normal, ndv), vec), n);
    float cos_sqr = 1.0f - (n * n) * (1.0f - ndv * ndv);
    if (cos_sqr >= 0.0f) { *out = vnormalize(vsub(s, vscale(normal, sqrtf(cos_sqr)))); return 1; }
    return 0;
}

typedef struct { uint32_t seed, pixel, sample; } rngctx;
static inline float rnd(const rngctx* r, uint32_t dim) { return orc_rng_f32(r->seed, r->pixel, r->sample, dim); }

static void smallest_axis_basis(v3 vec, v3* u, v3* v) {   /* scene.rs:395-404 / 414-423 */
    float xa = fabsf(vec.x), ya = fabsf(vec.y), za = fabsf(vec.z);
    v3 axis;
    if (xa < ya && xa < za) axis = V(1, 0, 0);
    else if (ya < za) axis = V(0, 1, 0);
    else axis = V(0, 0, 1);
    *u = vnormalize(vcross(vec, axis));
    *v = vnormalize(vcross(vec, *u));
}
/* scene.rs:394-411 */
static v3 random_rotation(v3 vec, float max_angle, float r0, float r1) {
    v3 u, v; smallest_axis_basis(vec, &u, &v);
    float z_min = cosf(max_angle);
    float z = z_min + r0 * (1.0f - z_min);
    float theta = r1 * 2.0f * PI_F;
    v3 o = vadd(vscale(vec, z), vscale(vadd(vscale(u, cosf(theta)), vscale(v, sinf(theta))), cosf(asinf(z))));
    return vnormalize(o);
}
/* scene.rs:413-429 */
static v3 random_hemisphere_sample(v3 vec, float r0, float r1) {
    v3 u, v; smallest_axis_basis(vec, &u, &v);
    float z = r0;
    float theta = r1 * 2.0f * PI_F;
    v3 o = vadd(vscale(vec, z), vscale(vadd(vscale(u, cosf(theta)), vscale(v, sinf(theta))), cosf(asinf(z))));
    return vnormalize(o);
}

static inline float powi2(float x) { return x * x; }
static inline float powi5(float x) { float x2 = x * x; return x2 * x2 * x; }  /* llvm powi: square-and-multiply */

/* scene.rs:130-188.  Random dimensions: 2 + 4*depth + {0: roulette, 1: lobe, 2: z, 3: theta}.
 * Deviation (shared with the GPU core, SURVEY 5-integrator "termination hazard"): a hit at depth ==
 * max_depth returns its emission only. */
static int cast(const orc_scene* s, v3 pos, v3 dir, float weight, int depth, int max_depth, const rngctx* rng, v3* out) {
    hitinfo hit; int32_t node = -1;
    memset(&hit, 0, sizeof hit);
    if (!scene_intersect(s, pos, dir, &hit, &node)) return 0;
    const orc_material* mat = &s->materials[s->nodes[node].material];
    v3 diffuse = texture_sample(s, &mat->diffuse, hit.uv);
    v3 reflection = texture_sample(s, &mat->reflection, hit.uv);
    v3 refraction = texture_sample(s, &mat->refraction, hit.uv);
    v3 normal = hit.side ? hit.normal : vneg(hit.normal);
    float n1, n2;
    if (hit.side) { n1 = 1.0f; n2 = mat->refraction_index; } else { n1 = mat->refraction_index; n2 = 1.0f; }
    float r0 = powi2((n1 - n2) / (n1 + n2));
    float ar = r0 + (1.0f - r0) * powi5(1.0f - vdot(normal, vneg(dir)));
    float p_diffuse = (diffuse.x + diffuse.y + diffuse.z) / 3.0f;
    float p_refraction = (1.0f - ar) * (refraction.x + refraction.y + refraction.z) / 3.0f;
    float p_reflection = (1.0f + ar) * (reflection.x + reflection.y + reflection.z) / 3.0f;
    float p_range = p_diffuse + p_reflection + p_refraction;
    v3 color = V(mat->emission[0], mat->emission[1], mat->emission[2]);
    *out = color;
    if (depth >= max_depth) return 1;
    uint32_t d0 = 2u + 4u * (uint32_t)depth;
    if (p_range == 0.0f || rnd(rng, d0) > weight) return 1;
    p_diffuse = p_diffuse / p_range;
    p_reflection = p_reflection / p_range;
    p_refraction = p_refraction / p_range;
    float r = rnd(rng, d0 + 1);
    v3 new_dir, albedo; float p;
    if (r < p_diffuse) {
        new_dir = random_hemisphere_sample(normal, rnd(rng, d0 + 2), rnd(rng, d0 + 3));
        albedo = diffuse; p = p_diffuse;
    } else if (r < p_diffuse + p_reflection) {
        new_dir = random_rotation(reflect_ray(vneg(dir), normal), mat->reflection_glossiness, rnd(rng, d0 + 2), rnd(rng, d0 + 3));
        albedo = reflection; p = p_reflection;
    } else if (r < p_diffuse + p_reflection + p_refraction) {
        v3 rd;
        if (!refract_ray(vneg(dir), normal, n1, n2, &rd)) rd = reflect_ray(vneg(dir), normal);
        new_dir = random_rotation(rd, mat->refraction_glossiness, rnd(rng, d0 + 2), rnd(rng, d0 + 3));
        albedo = refraction; p = p_refraction;
    } else return 1;
    v3 next;
    if (!cast(s, vadd(hit.pos, vscale(new_dir, 0.01f /* BIAS scene.rs:122 */)), new_dir, weight * p, depth + 1, max_depth, rng, &next))
        next = texture_sample_environment(s, &s->environment, new_dir);
    /* color += normal.dot(new_dir) * albedo.mul_element_wise(next) / p */
    color = vadd(color, vdivs(vscale(vmul(albedo, next), vdot(normal, new_dir)), p));
    *out = color;
    return 1;
}

/* main.rs:43-53 */
typedef struct { v3 a, right, up, pos; float pixel_width, pixel_height, dof; uint32_t w, h; } camframe;
static camframe cam_frame(const orc_camera* c) {
    camframe f;
    v3 dir = V(c->dir[0], c->dir[1], c->dir[2]);
    f.up = V(c->up[0], c->up[1], c->up[2]);
    f.pos = V(c->pos[0], c->pos[1], c->pos[2]);
    float height = tanf((c->fov / 2.0f) * (2.0f * PI_F / 360.0f)) * 2.0f * c->focaldist;
    float width = height * (float)c->img_width / (float)c->img_height;
    f.pixel_height = height / (float)c->img_height;
    f.pixel_width = width / (float)c->img_width;
    f.right = vcross(dir, f.up);
    v3 b = vadd(vadd(f.pos, vscale(dir, c->focaldist)), vscale(f.up, height / 2.0f));
    f.a = vsub(b, vscale(f.right, width / 2.0f));
    f.dof = c->dof; f.w = c->img_width; f.h = c->img_height;
    return f;
}
/* main.rs:65-85.  Lens randoms are dimensions 0 and 1. */
static void primary_ray(const camframe* f, uint32_t x, uint32_t y, int32_t i, const rngctx* rng, v3* eye, v3* dir, float* sx, float* sy) {
    v3 top_left = vsub(vadd(f->a, vscale(f->right, (float)x * f->pixel_width)), vscale(f->up, (float)y * f->pixel_height));
    float x_offset = orc_halton(i, 2);
    float y_offset = orc_halton(i, 3);
    float r1 = rnd(rng, 0), r2 = rnd(rng, 1);
    float eye_x = 2.0f * r1 - 1.0f, eye_y = 2.0f * r2 - 1.0f;
    v3 p = vsub(vadd(top_left, vscale(f->right, x_offset * f->pixel_width)), vscale(f->up, y_offset * f->pixel_height));
    *eye = vadd(vadd(f->pos, vscale(f->right, eye_x * f->dof)), vscale(f->up, eye_y * f->dof));
    *dir = vnormalize(vsub(p, *eye));
    *sx = ((float)x + x_offset) / (float)f->w;
    *sy = ((float)y + y_offset) / (float)f->h;
}
void orc_primary_ray(const orc_scene* s, uint32_t x, uint32_t y, int32_t i, uint32_t seed, float org[3], float dir[3]) {
    camframe f = cam_frame(&s->camera);
    rngctx rng = { seed, y * f.w + x, (uint32_t)i };
    v3 e, d; float sx, sy;
    primary_ray(&f, x, y, i, &rng, &e, &d, &sx, &sy);
    org[0] = e.x; org[1] = e.y; org[2] = e.z; dir[0] = d.x; dir[1] = d.y; dir[2] = d.z;
}

/* scene.rs:126-128 + main.rs:72-86 for one sample */
static v3 sample_pixel(const orc_scene* s, const camframe* f, uint32_t x, uint32_t y, int32_t i, uint32_t seed, int max_depth) {
    rngctx rng = { seed, y * f->w + x, (uint32_t)i };
    v3 eye, dir; float sx, sy;
    primary_ray(f, x, y, i, &rng, &eye, &dir, &sx, &sy);
    v3 c;
    if (!cast(s, eye, dir, 1.0f, 0, max_depth, &rng, &c)) c = texture_sample(s, &s->background, V(sx, sy, 0.0f));
    return c;
}

int orc_render(const orc_scene* s, const orc_render_params* p, float* rgb, uint32_t* samples_per_px,
               float* per_sample, orc_stats* stats) {
    camframe f = cam_frame(&s->camera);
    int64_t npix = (int64_t)f.w * f.h;
    int nt = pick_threads(p->threads);
    double t0 = now_s();
    orc_stats total; memset(&total, 0, sizeof total);
    int max_n = p->adaptive ? p->max_samples : p->spp;
    if (max_n <= 0) return -1;
#pragma omp parallel num_threads(nt)
    {
        orc_stats local; memset(&local, 0, sizeof local);
        tl_stats = &local;
        v3* samples = (v3*)malloc((size_t)max_n * sizeof(v3));
#pragma omp for schedule(dynamic, 64)
        for (int64_t i = 0; i < npix; i++) {
            uint32_t x = (uint32_t)(i % f.w), y = (uint32_t)(i / f.w);
            int32_t have = 0, num_samples;
            if (p->adaptive) {
                num_samples = p->initial_samples;
                for (;;) {   /* main.rs:71-107 */
                    for (int32_t k = have + 1; k <= num_samples; k++) samples[k - 1] = sample_pixel(s, &f, x, y, k, p->seed, p->max_depth);
                    have = num_samples;
                    float r_min = INFINITY, r_max = 0.0f, g_min = INFINITY, g_max = 0.0f, b_min = INFINITY, b_max = 0.0f;
                    for (int32_t k = 0; k < have; k++) {
                        if (samples[k].x < r_min) r_min = samples[k].x;
                        if (samples[k].x > r_max) r_max = samples[k].x;
                        if (samples[k].y < g_min) g_min = samples[k].y;
                        if (samples[k].y > g_max) g_max = samples[k].y;
                        if (samples[k].z < b_min) b_min = samples[k].z;
                        if (samples[k].z > b_max) b_max = samples[k].z;
                    }
                    if (num_samples < p->max_samples &&
                        ((r_max - r_min) / (r_min + r_max) > p->thresholds[0] || (g_max - g_min) / (g_min + g_max) > p->thresholds[1] ||
                         (b_max - b_min) / (b_min + b_max) > p->thresholds[2])) {
                        num_samples *= 2;
                    } else break;
                }
            } else {
                num_samples = p->spp;
                for (int32_t k = 1; k <= num_samples; k++) samples[k - 1] = sample_pixel(s, &f, x, y, p->first_sample + k, p->seed, p->max_depth);
                have = num_samples;
            }
            v3 tot = V(0, 0, 0);   /* main.rs:109: iter().sum() from zero */
            for (int32_t k = 0; k < have; k++) tot = vadd(tot, samples[k]);
            v3 c = vdivs(tot, (float)num_samples);
            rgb[3 * i] = c.x; rgb[3 * i + 1] = c.y; rgb[3 * i + 2] = c.z;
            if (samples_per_px) samples_per_px[i] = (uint32_t)num_samples;
            if (per_sample && !p->adaptive)
                for (int32_t k = 0; k < have; k++) {
                    float* d = &per_sample[((size_t)i * (size_t)p->spp + (size_t)k) * 3];
                    d[0] = samples[k].x; d[1] = samples[k].y; d[2] = samples[k].z;
                }
        }
        free(samples);
        tl_stats = NULL;
#pragma omp critical
        stats_merge(&total, &local);
    }
    if (stats) { *stats = total; stats->seconds = now_s() - t0; stats->threads = nt; }
    return 0;
}
