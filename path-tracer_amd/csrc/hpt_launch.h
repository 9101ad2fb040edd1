// hpt_launch.h -- host-callable launchers of the kernels in hpt_kernels.hip (all asynchronous on `st`).
#pragma once
#include <hip/hip_runtime.h>
#include "hpt_device.h"

size_t hptk_mq_slots(uint64_t n);      // items a sharded mesh queue for n rays holds (test entries; a pass uses its queue geometry)
void hptk_trace_rays(hipStream_t st, int max_blocks, const DScene& sc, const DMesh& mesh, long long n, const float4* org,
                     const float4* dir, float4* hit, uint8_t* side, bool count, unsigned long long* stat, uint2* mq, uint32_t* mq_count);
void hptk_scene_intersect(hipStream_t st, int max_blocks, int stack_entries, const DScene& sc, long long n, const float4* org,
                          const float4* dir, float4* hit, uint32_t* hit_inst, int* node, int* prim, float* t, uint8_t* side,
                          float4* pos, float4* normal, float4* uvw, uint2* mq, uint32_t* mq_count);
void hptk_raygen(hipStream_t st, int max_blocks, const DScene& sc, const DPass& ps, bool fuse, bool count);
struct HptkPlan {          // what hptk_plan derives from a committed scene (see hpt_kernels.hip)
    int extend_entries = 1, extend_mode = 0, extend_per_cu = 0;   // stack entries carved, LDS layout, resident workgroups per CU (0: query failed)
    size_t extend_lds = 0;
    bool extend_counter = false;                                  // the traversal kernel takes its input through the work counter (two-level scenes, big trees)
    int analytic_per_cu = 0;                                      // resident workgroups per CU of k_extend_a
    size_t shade_tables = 0; bool shade_lds = false; int shade_per_cu = 0, shade_fused_per_cu = 0;   // (fused: k_shade + the next depth's analytic stage)
};
void hptk_plan(int stack_entries, const DScene& sc, int n_xf, int n_mat, HptkPlan* out);
// instance-list scenes: analytic stage of a depth (every ray; fills the mesh queue), then hptk_extend over the mesh queue
void hptk_extend_analytic(hipStream_t st, int blocks, const DScene& sc, const DPass& ps, int depth, bool count);
// big trees (hptk_wants_binning): mesh queue of a depth >= 1 copied into cell order (DPass.mq_bin) before hptk_extend walks it
bool hptk_wants_binning(const DScene& sc);
void hptk_mq_bin(hipStream_t st, int blocks, const DScene& sc, const DPass& ps, int depth);
void hptk_extend(hipStream_t st, int blocks, const HptkPlan& plan, const DScene& sc, const DPass& ps, int depth, bool count);
void hptk_shade(hipStream_t st, int blocks, const HptkPlan& plan, const DScene& sc, const DPass& ps, int depth, int n_xf, int n_mat, bool fuse, bool count, bool exact);
void hptk_halton(hipStream_t st, float2* out, uint32_t first_sample, uint32_t n);      // out[k] = halton(first_sample + k + 1, 2 | 3)
void hptk_resolve(hipStream_t st, int max_blocks, const DPass& ps, long long* accum, float* minmax);
void hptk_finish_pass(hipStream_t st, const DPass& ps);
void hptk_adaptive_select(hipStream_t st, int max_blocks, const uint32_t* in_list, uint32_t n_in, const float* minmax, const float thr[3],
                          uint32_t* out_list, uint32_t* out_count, uint32_t* samples_per_px, uint32_t num_samples_next);
void hptk_fill_f32(hipStream_t st, int max_blocks, float* p, size_t n, float v);
// group render (hpt_group.cpp): partial results of the ranks combined on one device
void hptk_add_i64(hipStream_t st, int max_blocks, long long* dst, const long long* src, size_t n);
void hptk_minmax_merge(hipStream_t st, int max_blocks, float* dst, const float* src, size_t n);
void hptk_negate_min(hipStream_t st, int max_blocks, float* p, size_t n);
void hptk_init_minmax(hipStream_t st, int max_blocks, float* p, uint32_t npix);      // main.rs:90-92
// this device's address of the static workspace of the no-alloc entry point (hipGetSymbolAddress, once per device; the
// device must be current).  0 on success.
int hptk_scratch(int device, void** out, size_t* bytes);
void hptk_fill_u32(hipStream_t st, int max_blocks, uint32_t* p, size_t n, uint32_t v);
void hptk_accum_resolve(hipStream_t st, int max_blocks, const long long* accum, const uint32_t* samples, float* out, uint32_t npix, float fixed_n);
void hptk_set_refill(int lanes);   // tuning: idle lanes per wave that trigger a refill in drain_queue
