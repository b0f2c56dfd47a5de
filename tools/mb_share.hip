// Microbenchmark (design probe, not product): can the row pass at 16384^2 take the last radix-4 step of the x transform
// (nx = 4 x 4096, DESIGN.md section 7 "Next (1)") if the four workgroups that need the same four half-transformed rows
// Y_k1[x2] (k1 = 0..3, 64 KB each, per field) run on ONE XCD at the same time, so that three of the four reads hit that XCD's L2?
//   mode 0: today's traffic -- workgroup x reads row x of each of the four fields (prefetched one field ahead), writes row x of T
//   mode 1: team of four on one XCD (blocks b, b+8, b+16, b+24): workgroup (x2, x1) reads rows k1*4096 + x2, k1 = 0..3, of each field
//   mode 2: as 1, but the four members of a team are 16 blocks apart in launch order... on four different XCDs (b, b+1, b+2, b+3)
// 512 threads and 140 KB of LDS per workgroup (one per CU, as k_rowh<2>); a field phase is DELAY x s_sleep(127) (~3.4 us each at
// 2.4 GHz) standing in for the transform, with the next field's loads in flight meanwhile.
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/mb_share tools/mb_share.hip ; run under rocprofv3 --pmc FETCH_SIZE for the L2 misses.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#ifndef MB_NT
#define MB_NT 0
#endif
typedef float f4v __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(512) k_rows(const float2 *__restrict__ w4, float2 *__restrict__ tout, long fstride, int P, int nx, int delay)
{
    extern __shared__ unsigned char smem[];
    const int b = blockIdx.x, t = threadIdx.x;
    int x2, x1;
    if (MODE == 1) { const int xcd = b & 7, j = b >> 3; x1 = j & 3; x2 = (j >> 2) * 8 + xcd; }
    else if (MODE == 2) { x1 = b & 3; x2 = b >> 2; }
    else { x1 = b / (nx / 4); x2 = b - x1 * (nx / 4); }
    const int xrow = x1 * (nx / 4) + x2;
    constexpr int NR = MODE == 0 ? 1 : 4;
    f4v cur[NR][8], acc = {0.f, 0.f, 0.f, 0.f};
    auto issue = [&](int f, f4v (*dst)[8]) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int row = MODE == 0 ? xrow : r * (nx / 4) + x2;
            const f4v *src = reinterpret_cast<const f4v *>(w4 + (size_t)f * fstride + (size_t)row * P) + t;
#pragma unroll
            for (int c = 0; c < 8; ++c) dst[r][c] = MB_NT ? __builtin_nontemporal_load(src + c * 512) : src[c * 512];
        }
    };
    issue(0, cur);
    for (int f = 0; f < 4; ++f) {
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int c = 0; c < 8; ++c) acc += cur[r][c];
        if (f + 1 < 4) issue(f + 1, cur);                              // next field's rows travel during the "transform"
        for (int d = 0; d < delay; ++d) __builtin_amdgcn_s_sleep(127);
        __syncthreads();
    }
    if (t == 0) smem[0] = (unsigned char)acc.x;
    f4v *dst = reinterpret_cast<f4v *>(tout + (size_t)xrow * P) + t;
#pragma unroll
    for (int c = 0; c < 8; ++c) dst[c * 512] = acc + (float)c;
}

int main(int argc, char **argv)
{
    const int nx = 16384, P = 8208;
    const int delay = argc > 1 ? atoi(argv[1]) : 1, reps = argc > 2 ? atoi(argv[2]) : 5;
    const size_t n = (size_t)nx * P;
    float2 *w4, *tt;
    CK(hipMalloc(&w4, 4 * n * sizeof(float2)));
    CK(hipMalloc(&tt, n * sizeof(float2)));
    CK(hipMemset(w4, 0, 4 * n * sizeof(float2)));
    CK(hipMemset(tt, 0, n * sizeof(float2)));
    const size_t lds = 140 * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_rows<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_rows<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_rows<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double unique = 5.0 * nx * 8192.0 * 8;
#define RUN(MODE) { \
        for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k_rows<MODE>), dim3(nx), dim3(512), lds, 0, w4, tt, (long)n, P, nx, delay); \
        hipDeviceSynchronize(); hipEventRecord(e0, 0); \
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_rows<MODE>), dim3(nx), dim3(512), lds, 0, w4, tt, (long)n, P, nx, delay); \
        hipEventRecord(e1, 0); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps; \
        printf("mode %d delay %d : %.3f ms per launch, %.0f GB/s of unique bytes\n", MODE, delay, ms, unique / ms / 1e6); }
    RUN(0) RUN(1) RUN(2)
    CK(hipDeviceSynchronize());
    return 0;
}
