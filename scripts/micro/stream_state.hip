// What does the memory system give a kernel with k_shade's access pattern and NO arithmetic?  Per path: three 16-B queue records +
// a 16-B hit record + a 4-B instance word read at consecutive slots (68 B), three 16-B records written compacted (ballot + prefix,
// one atomic per wave) for the paths that survive (48 B x survival).  k_shade runs at ~23-26 ps per path on the 1080p teapot frame;
// this is the floor its byte count sets.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/micro/stream_state scripts/micro/stream_state.hip
// Variants: R arrays read (5 = k_shade; 3 = without the hit records), W arrays written (3 / 0), survival in 1/16ths, grid shape
// (one thread per path, or a persistent grid of 14 workgroups per CU striding over 256-path pieces like k_shade).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
struct Arrays { const float4 *a, *b, *c, *h; const uint32_t* hi; float4 *x, *y, *z; uint32_t* count; };
template <int R, int W>
__global__ void __launch_bounds__(256) k_state(Arrays A, uint32_t n, uint32_t keep16) {
    const unsigned lane = threadIdx.x & 63u;
    const uint32_t pieces = (n + 255u) / 256u;
    for (uint32_t w = blockIdx.x; w < pieces; w += gridDim.x) {
        const uint32_t i = w * 256u + threadIdx.x;
        bool alive = false;
        float4 o = make_float4(0, 0, 0, 0), d = o, t = o;
        if (i < n) {
            o = A.a[i]; d = A.b[i]; t = A.c[i];
            if (R >= 4) { const float4 h = A.h[i]; o.x += h.x; d.y += h.y; t.z += h.z + h.w; }
            if (R >= 5) { const uint32_t s = A.hi[i]; o.w += (float)s; }
            alive = ((i * 2654435761u) >> 28) < keep16;
        }
        if (W == 0) { if (o.x + d.y + t.z + o.w == 123.456f) A.count[1] = 1; continue; }
        const unsigned long long mask = __ballot(alive);
        if (mask) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&A.count[(w & 63u) * 16u], (uint32_t)__popcll(mask));
            base = __shfl(base, 0);
            if (alive) {
                // segment (w & 63) of the output, like the sharded queues: survivors of a piece stay together
                const uint32_t dst = (w & 63u) * (((pieces + 63u) / 64u) * 256u) + base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
                A.x[dst] = o; A.y[dst] = d; A.z[dst] = t;
            }
        }
    }
}
template <int R, int W>
static void run(const char* what, Arrays A, uint32_t n, uint32_t keep16, int grid) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipMemset(A.count, 0, 64 * 16 * 4 + 64));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_state<R, W>), dim3(grid), dim3(256), 0, 0, A, n, keep16);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double bytes = (double)n * ((R >= 3 ? 48 : 0) + (R >= 4 ? 16 : 0) + (R >= 5 ? 4 : 0) + W * 16 * keep16 / 16.0);
        if (rep) std::printf("%-46s grid %6d  %7.3f ms  %5.1f ps per path  %6.0f GB/s of %5.1f B per path\n", what, grid, ms, ms * 1e9 / n, bytes / ms / 1e6, bytes / n);
    }
}
int main() {
    const uint32_t n = 62u << 20;      // the teapot frame's mean launch
    const size_t slots = (size_t)n + 64 * 256 * 2;
    Arrays A; float4* p[7];
    for (int k = 0; k < 7; ++k) { CHECK(hipMalloc(&p[k], slots * 16)); CHECK(hipMemset(p[k], 0, slots * 16)); }
    uint32_t* hi; CHECK(hipMalloc(&hi, slots * 4)); CHECK(hipMemset(hi, 0, slots * 4));
    uint32_t* count; CHECK(hipMalloc(&count, 64 * 16 * 4 + 64));
    A.a = p[0]; A.b = p[1]; A.c = p[2]; A.h = p[3]; A.hi = hi; A.x = p[4]; A.y = p[5]; A.z = p[6]; A.count = count;
    const int flat = (int)((n + 255u) / 256u), pers = 256 * 14;
    run<5, 3>("read 68 B, write 48 B x 14/16  (k_shade)", A, n, 14, pers);
    run<5, 3>("read 68 B, write 48 B x 14/16  (k_shade)", A, n, 14, flat);
    run<5, 3>("read 68 B, write 48 B x 16/16", A, n, 16, pers);
    run<5, 0>("read 68 B only", A, n, 14, pers);
    run<3, 0>("read 48 B only", A, n, 14, pers);
    run<3, 3>("read 48 B, write 48 B x 14/16", A, n, 14, pers);
    run<3, 3>("read 48 B, write 48 B x 14/16", A, n, 14, 256 * 28);
    run<3, 3>("read 48 B, write 48 B x 14/16", A, n, 14, 256 * 8);
    return 0;
}
