// What does rocprofv3's FETCH_SIZE report for the access pattern of the BVH traversal -- every lane reading a 32-byte record (two
// dwordx4) at a random place of a table that does not fit the L2s -- compared with the bytes the lanes really asked for?
// MI355X_MICROARCH.md calibrates the counter for wide coalesced streaming reads only (gfx950 tallies them at half their bytes) and
// says other access widths are uncalibrated.  VERDICT r02 weak 5: profiles/r02g_soup_pmc_traffic.json doubled FETCH_SIZE for k_extend
// on the 1 M-triangle soup (112 MB of Node32 records + triangles, gathered) without that calibration.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/micro/gather32 scripts/micro/gather32.hip
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -o p -- scripts/micro/gather32        (then scripts/pmc_summary.py out)
// Kernels (template tag = table size in MB): k_gather<MB> random 32-B records, k_gather64<MB> random 64-B records (one Node64),
// k_stream<MB> the same loads at consecutive records (coalesced: 32 B per lane, 2 KB per wave).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
__device__ __forceinline__ uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
template <int MB, int REC16, bool RANDOM>      // REC16: 16-byte quads per record (2 = 32 B, 4 = 64 B)
__global__ void __launch_bounds__(256) k_read(const uint4* __restrict__ table, uint32_t n_records, int iters, uint32_t* __restrict__ out) {
    const uint32_t tid = blockIdx.x * 256u + threadIdx.x, nthreads = gridDim.x * 256u;
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
        const uint32_t k = RANDOM ? mix32(tid * 2654435761u + (uint32_t)i * 40503u + 12345u) % n_records : (tid + (uint32_t)i * nthreads) % n_records;
        const uint4* r = table + (size_t)k * REC16;
#pragma unroll
        for (int q = 0; q < REC16; ++q) { const uint4 v = r[q]; acc ^= v.x + v.y + v.z + v.w; }
    }
    out[tid] = acc;
}
template <int MB, int REC16, bool RANDOM>
static void run(const char* what, const uint4* table, uint32_t* out, int grid, int iters) {
    const uint32_t n_records = (uint32_t)(((size_t)MB << 20) / (16 * REC16));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_read<MB, REC16, RANDOM>), dim3(grid), dim3(256), 0, 0, table, n_records, iters, out);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double bytes = (double)grid * 256 * iters * 16 * REC16;
        std::printf("%-34s table %5d MB  requested %8.1f MB per launch  %7.3f ms  %7.1f GB/s\n", what, MB, bytes / 1e6, ms, bytes / ms / 1e6);
    }
}
int main() {
    const size_t big = (size_t)1024 << 20;
    uint4* table; uint32_t* out;
    CHECK(hipMalloc(&table, big)); CHECK(hipMemset(table, 1, big));
    const int grid = 256 * 16, iters = 64;      // 1 M lanes x 64 records: 2.1 GB of 32-B records per launch
    CHECK(hipMalloc(&out, (size_t)grid * 256 * 4));
    run<128, 2, true>("k_read<128,2,true>  random 32 B", table, out, grid, iters);
    run<1024, 2, true>("k_read<1024,2,true> random 32 B", table, out, grid, iters);
    run<128, 4, true>("k_read<128,4,true>  random 64 B", table, out, grid, iters);
    run<1024, 4, true>("k_read<1024,4,true> random 64 B", table, out, grid, iters);
    run<128, 2, false>("k_read<128,2,false> stream 32 B/lane", table, out, grid, iters);
    run<1024, 2, false>("k_read<1024,2,false> stream 32 B/lane", table, out, grid, iters);
    CHECK(hipFree(table)); CHECK(hipFree(out));
    return 0;
}
