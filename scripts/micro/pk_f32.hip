// Is packed f32 arithmetic (v_pk_mul_f32 / v_pk_add_f32: two IEEE operations per lane and instruction) twice as fast as
// scalar-per-lane f32 on gfx950 when a kernel is bound by VALU issue?  Same number of multiply / add OPERATIONS both ways.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <bool PACKED>
__global__ void __launch_bounds__(256) k(float* out, float a, float b, int iters) {
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, x4 = x0 + 4.f, x5 = x0 + 5.f, x6 = x0 + 6.f, x7 = x0 + 7.f;
    if (PACKED) {
        v2f p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
        const v2f va = {a, a}, vb = {b, b};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) { p0 = p0 * va; p0 = p0 + vb; p1 = p1 * va; p1 = p1 + vb; p2 = p2 * va; p2 = p2 + vb; p3 = p3 * va; p3 = p3 + vb; }
        }
        x0 = p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    } else {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                x0 = x0 * a; x0 = x0 + b; x1 = x1 * a; x1 = x1 + b; x2 = x2 * a; x2 = x2 + b; x3 = x3 * a; x3 = x3 + b;
                x4 = x4 * a; x4 = x4 + b; x5 = x5 * a; x5 = x5 + b; x6 = x6 * a; x6 = x6 + b; x7 = x7 * a; x7 = x7 + b;
            }
        }
        x0 = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0;
}
int main() {
    float* d; hipMalloc(&d, 256 * 256 * 8 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000, grid = 256 * 8;
    for (int rep = 0; rep < 2; ++rep)
        for (int packed = 0; packed < 2; ++packed) {
            hipEventRecord(e0);
            if (packed) hipLaunchKernelGGL(k<true>, dim3(grid), dim3(256), 0, 0, d, 0.999f, 0.001f, iters);
            else hipLaunchKernelGGL(k<false>, dim3(grid), dim3(256), 0, 0, d, 0.999f, 0.001f, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double ops = (double)grid * 256 * iters * 8 * 16;      // f32 multiply / add operations
            printf("%s: %.3f ms, %.1f Tops/s\n", packed ? "packed (v_pk_*_f32)" : "plain  (v_*_f32)  ", ms, ops / ms / 1e9);
        }
    return 0;
}
