// hpt_device.h -- data layout in HBM shared by the host-side flattener and the gfx950 kernels.
//
// Everything the kernels read is plain float4 / int arrays ("SoA of 16-byte quads"):
//   Node64 : 4 x float4 per interior BVH2 node = one 64-byte record holding BOTH child boxes and child refs
//            q0 = (left.min.xyz,  as_float(left_ref))   q1 = (left.max.xyz,  as_float(right_ref))
//            q2 = (right.min.xyz, 0)                    q3 = (right.max.xyz, 0)
//            child ref >= 0: index of an interior node (global over all meshes), < 0: ~(global triangle index).
//            Leaves (bvh.rs:19-22) carry exactly one primitive, so they are folded into their parent's refs.
//            Interior nodes of one mesh are stored in breadth-first order (top levels first => cacheable prefix).
//   Node32 : the same record with 16-bit child boxes on a per-tree grid, 2 x uint4 = 32 bytes (default when the grid
//            resolves the leaves; halves the per-lane fetches of a traversal step -- the texture addresser, not bytes,
//            limits the divergent node fetch):
//            q0 = (Lx, Ly, Lz, left_ref)  q1 = (Rx, Ry, Rz, right_ref), box word = min | max << 16, coordinate =
//            qbase + q * qscale (DMesh / DScene.tlas_*); boxes are rounded outward, so culling stays conservative.
//   Tri48  : 3 x float4 per triangle: (a.xyz,0) (b-a,0) (c-a,0)  -- the operands of geometry.rs:188-199
//   TriShade: 7 x float4 per triangle: a,b,c (get_point), n0,n1,n2 (get_normal), t0,t1,t2 (get_texture_vertex),
//            27 floats + 1 pad, fetched once per closest hit by the shade stage only.
#pragma once
#include <stdint.h>

#ifndef HPT_BLOCK
#define HPT_BLOCK 256          // threads per workgroup: 4 wave64 (128 and 512 measured slower)
#endif
#define HPT_NODE_QUADS 4
#ifndef HPT_TRI_QUADS
#define HPT_TRI_QUADS 4          // 16-B quads per triangle record: (a, b - a, c - a) = 48 B, padded to 64 B so that no record straddles a 128-B line
                                 // (one in four 48-B records does; r03g A/B of 3 vs 4: soup +1.3 %, k_extend -1.6 %; teapot, instanced teapots +-0)
#endif
#ifndef HPT_TRI_FLOATS
#define HPT_TRI_FLOATS (HPT_TRI_QUADS * 4)
#endif
#define HPT_TRISHADE_QUADS 7
#define HPT_MISS_INST 0x7fffffffu
#define HPT_MAX_DEPTH_LIMIT 64   // max bounces (size of the per-depth live-count array)
#ifndef HPT_SHARDS
#define HPT_SHARDS 64
#endif
//      HPT_SHARDS               // segments per wavefront queue (one append counter each)
#ifndef HPT_COUNT_STRIDE
#define HPT_COUNT_STRIDE 32      // uint32 slots between two append counters: one counter per 128-B line (L2 atomics to one line serialise)
#endif
#define HPT_COUNTS_PER_DEPTH (HPT_SHARDS * HPT_COUNT_STRIDE)
// DPass.counts holds HPT_MAX_DEPTH_LIMIT + 2 blocks of append counters, then one work counter per depth (k_extend hands
// its input out in batches of HPT_PIECE_BATCH 64-ray pieces), HPT_COUNT_STRIDE apart like the others
// ... then one block of append counters per depth for the MESH QUEUE (k_extend_a appends the rays that enter a mesh box,
// k_extend walks them), laid out like the path-queue counters
#define HPT_WORK_COUNTER(depth) ((HPT_MAX_DEPTH_LIMIT + 2) * HPT_COUNTS_PER_DEPTH + (depth) * HPT_COUNT_STRIDE)
#define HPT_MQ_COUNTS(depth) ((HPT_MAX_DEPTH_LIMIT + 2) * (HPT_COUNTS_PER_DEPTH + HPT_COUNT_STRIDE) + (depth) * HPT_COUNTS_PER_DEPTH)
#define HPT_COUNTS_TOTAL ((HPT_MAX_DEPTH_LIMIT + 2) * (2 * HPT_COUNTS_PER_DEPTH + HPT_COUNT_STRIDE))
// binned mesh queue (big trees, k_mq_hist / k_mq_scatter): per depth a 64-cell histogram, 64 scatter cursors and the total, in the unused words of the
// 128-B lines of that depth's mesh-queue append counters (cell c: words 1 and 2 of line c; total: word 3 of line 0)
#define HPT_MQ_CELLS 64
#define HPT_MQ_HIST(depth, cell) (HPT_MQ_COUNTS(depth) + (cell) * HPT_COUNT_STRIDE + 1)
#define HPT_MQ_CURSOR(depth, cell) (HPT_MQ_COUNTS(depth) + (cell) * HPT_COUNT_STRIDE + 2)
#define HPT_MQ_BINNED_TOTAL(depth) (HPT_MQ_COUNTS(depth) + 3)
#ifndef HPT_PIECE_BATCH
#define HPT_PIECE_BATCH 8
#endif

// ---- the radiance accumulator (include/hpt.h hpt_accum): 4 x int64 per pixel, sums in fixed point ----
#define HPT_ACCUM_SCALE_F 16777216.0f          // 2^HPT_ACCUM_FRAC_BITS
#define HPT_ACCUM_SCALE_D 16777216.0
#define HPT_ACCUM_SAMPLE_MAX_F 1048576.0f      // HPT_ACCUM_SAMPLE_MAX

// ---- static workspace of hpt_render_device_noalloc (hpt_kernels.hip `scratch`) ----
#define HPT_NOALLOC_MAX_PATHS (1u << 19)   // pixels of the largest frame it renders (one sample of every pixel per pass; the reference's 800 x 600 fits)
#ifndef HPT_NOALLOC_MAX_SPP
#define HPT_NOALLOC_MAX_SPP 4096           // samples per call (size of the Halton table); include/hpt.h states the same limits
#endif
// queue slots: 64 segments, each rounded up to whole 256-path chunks
#define HPT_NOALLOC_SLOTS ((size_t)HPT_NOALLOC_MAX_PATHS + (size_t)HPT_SHARDS * HPT_BLOCK)
#define HPT_ALIGN256(x) (((size_t)(x) + 255) & ~(size_t)255)
// regions, each 256-B aligned: 6 queue arrays + 2 x hit + radiance (16 B per slot), 2 x hit_inst (4 B), mesh queue (3 x 16 B), counters, statistics, Halton,
// the radiance accumulator of the call (32 B per pixel)
#define HPT_NOALLOC_OFF_Q(k) ((size_t)(k) * HPT_ALIGN256(HPT_NOALLOC_SLOTS * 16))
#define HPT_NOALLOC_OFF_HIT HPT_NOALLOC_OFF_Q(6)            // two arrays (even / odd depths: DPass.hit_par = HPT_NOALLOC_SLOTS)
#define HPT_NOALLOC_OFF_RADIANCE HPT_NOALLOC_OFF_Q(8)
#define HPT_NOALLOC_OFF_HIT_INST HPT_NOALLOC_OFF_Q(9)       // two arrays as well
#define HPT_NOALLOC_OFF_MQ (HPT_NOALLOC_OFF_HIT_INST + HPT_ALIGN256(HPT_NOALLOC_SLOTS * 8))
#define HPT_NOALLOC_OFF_COUNTS (HPT_NOALLOC_OFF_MQ + HPT_ALIGN256(HPT_NOALLOC_SLOTS * 48))
#define HPT_NOALLOC_OFF_STAT (HPT_NOALLOC_OFF_COUNTS + HPT_ALIGN256((size_t)HPT_COUNTS_TOTAL * 4))
#define HPT_NOALLOC_OFF_HALTON (HPT_NOALLOC_OFF_STAT + 256)
#define HPT_NOALLOC_OFF_ACCUM (HPT_NOALLOC_OFF_HALTON + HPT_ALIGN256((size_t)HPT_NOALLOC_MAX_SPP * 8))
#define HPT_NOALLOC_SCRATCH_BYTES (HPT_NOALLOC_OFF_ACCUM + HPT_ALIGN256((size_t)HPT_NOALLOC_MAX_PATHS * 32))

struct DMesh {
    int32_t root_ref;     // ref of the root (interior index or ~tri for a one-triangle mesh)
    int32_t tri_base;     // global index of this mesh's triangle 0
    int32_t ntri;
    int32_t stack_need;   // traversal stack entries needed (tree depth + 1)
    float bmin[3], bmax[3];   // tight root box
    float qbase[3], qscale[3];   // Node32 grid of this mesh's tree
};

// one level of the Node chain of an instance (scene.rs:102-105 Transform), rows of 3x3 matrices as float4
struct DXformInv {   // extend: to_local(v) = Minv * (v - t)       (scene.rs:278-280)
    float r0[4], r1[4], r2[4];   // (Minv row r, t[r])
};
struct DXformFwd {   // shade: from_local(v) = M*v + t (scene.rs:282), normal = normalize(Minv^T n) (scene.rs:286)
    float m0[4], m1[4], m2[4];   // (M row r, t[r])
    float n0[4], n1[4], n2[4];   // (Minv^T row r, 0)
};

struct DInstance {   // one Node with an object, in the reference's DFS visiting order.  128 bytes, self-contained:
    int32_t geom, mesh, material, node_id;
    int32_t xf_first, xf_count;   // chain in xinv[]/xfwd[]: outermost ancestor first
    int32_t shared_top;           // 1: level 0 is the same Node as the previous instance's level 0 (reuse its local ray)
    int32_t pad;
    DXformInv lv[2];              // inline copy of chain levels 0 and 1 (deeper levels: xinv[xf_first + k])
};

struct DTexture {
    int32_t kind, image;
    float color[3], color1[3], color2[3];
    float minv[9];    // row-major inverse of the texture matrix (r*3+c)
    float t[3];
};
struct DMaterial {
    DTexture diffuse, reflection, refraction;
    float emission[3];
    float refl_gloss, refr_gloss, ior;
    // per-material constants of the bounce, computed once at commit with the host's IEEE arithmetic / libm (the values
    // scene.rs:150,408 recompute per hit): cos of the two glossiness cone angles, Schlick r0 = ((1 - ior) / (1 + ior))^2
    float cos_refl_gloss, cos_refr_gloss, r0;
};
struct DImage {
    uint64_t offset;   // byte offset into the image pool
    int32_t w, h;
};
struct DCamera {      // main.rs:43-53 frame, computed on the host in the reference's f32 order
    float a[3], right[3], up[3], pos[3];
    float pixel_width, pixel_height, dof;
    uint32_t w, h;
};

struct DScene {
    const float4* nodes;
    const float4* tris;
    const float4* trishade;
    const DMesh* meshes;
    const DInstance* instances;
    const int32_t* mesh_instances;   // indices (into instances[]) of the mesh instances, ascending
    const DXformInv* xinv;
    const DXformFwd* xfwd;
    const float4* plane_normal;      // per instance: world-space normal of a plane instance (0 otherwise); hpt_scene_commit
    const DMaterial* materials;
    const DImage* images;
    const uint8_t* image_pool;
    int32_t n_instances, n_meshes, n_mesh_instances, n_nodes;
    int32_t use_tlas, tlas_root;     // two-level traversal (many instances): ref of the TLAS root in nodes[]
    int32_t node32, stk16_ok;        // node32 = 1: nodes[] holds Node32 records (2 quads per node), 0: Node64 (4 quads); stk16_ok: see inst_base
    // leaf codes of the two-level path: a leaf ref is ~code; code < inst_base = triangle (global index), < analytic_base = mesh instance
    // (inst_base + i), else sphere / plane (analytic_base + i); restore_ref = the marker a ray pushes when it enters a mesh instance.
    // 0x40000000 / 0x60000000 / ~0x7ffffffe -- or, when triangles + 2 x instances + the marker number below 32 767 and there are fewer
    // than 32 767 interior nodes, packed from 0 (stk16_ok = 1: every stack entry fits a signed 16-bit word)
    int32_t inst_base, analytic_base, restore_ref;
    int32_t tri_defer;               // two-level path: deferred triangle tests (HPT_TLAS_TRI_DEFER=1 at commit; A/B variant, see hpt_kernels.hip tlas_meshes)
    float tlas_qbase[3], tlas_qscale[3];   // Node32 grid of the TLAS
    DTexture background, environment;
    DCamera cam;
};

// wavefront queues (SoA).  A path keeps its identity (path id = pixel_slot * S + s) in ray_d.w.
struct DQueue {
    float4* ray_o;   // (origin.xyz, roulette weight)                       scene.rs:130 `weight`
    float4* ray_d;   // (dir.xyz, as_float(index of the path's radiance record))
    float4* thr;     // (throughput.rgb, 0): product of n.dot(new_dir)*albedo/p over the bounces so far
};

struct DPass {
    DQueue q[2];
    float4* hit;         // (t, u, v, as_float(global tri or -1))
    uint32_t* hit_inst;  // instance | side<<31, HPT_MISS_INST on miss
    uint32_t hit_par;    // records between the hit / hit_inst arrays of even and odd depths (fused shade + analytic stage: k_shade of depth d writes
                         // the analytic results of depth d + 1 while depth d's records are still read); 0: one array for every depth
    uint4* mq;           // mesh queue of the current depth, sharded like the path queues: for every ray that entered a mesh box the item
                         // (slot, first mesh instance to visit, best analytic t, its instance | side), then -- HPT_SHARDS * shard_cap records
                         // further on each -- the ray's origin and direction (48 B per slot; see block_analytic)
    uint4* mq_bin;       // NULL, or a second mesh queue of the same size: scenes with big trees (beyond the L2s) walk the rays of depths >= 1 in the
                         // order of the 4 x 4 x 4 cell of the first mesh's box their origin lies in (k_mq_hist + k_mq_scatter fill it as ONE linear
                         // array; its length sits at counts[HPT_MQ_BINNED_TOTAL(depth)])
    float4* radiance;    // (L.rgb, 0) of sample s of slot i at [s * n_slots + i] (k_resolve reads it coalesced); zeroed by the pass
    uint32_t* counts;    // [HPT_MAX_DEPTH_LIMIT+2][HPT_SHARDS] live paths per depth and queue segment, HPT_COUNT_STRIDE apart
    uint32_t shard_cap;  // slots per segment (multiple of HPT_BLOCK); a queue holds HPT_SHARDS * shard_cap slots
    const float2* halton;        // [S] (halton(i,2), halton(i,3)) of this pass's samples
    const uint32_t* pixel_list;  // adaptive: active pixels of this pass (NULL = all pixels, slot == pixel)
    uint32_t n_slots;            // pixels in this pass
    uint32_t S;                  // samples per pixel in this pass
    uint32_t first_sample;       // Halton index of sample s is first_sample + s + 1
    uint32_t seed;
    int32_t max_depth;
    uint32_t flags;              // HPT_RENDER_* of the call (k_shade launch: HPT_RENDER_EXACT_SHADE)
    unsigned long long* stat;    // [0] rays, [1] nodes, [2] tris, [3] inst tests, [4] primary rays, [5] rays that entered a mesh box
};
