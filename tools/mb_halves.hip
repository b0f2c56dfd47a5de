// Microbenchmark (design probe, not product): what does it cost to write the 64-byte row segments of k_col_full's four derivative
// fields in two 32-byte halves that leave ~10 us apart ("Next (0)": two fields of one column set per pass)
// instead of whole?  One 1024-thread workgroup per 8-column tile of a [4096][2096] complex64 array, four fields, stores only.
//   full  : 4 passes, pass f writes field f, 16 B per lane (columns 2c, 2c+1)
//   halves: 4 passes, pass p writes fields 2(p/2), 2(p/2)+1, column set p%2 (columns c + 4 (p%2)), 8 B per lane
// A pass is followed by DELAY x s_sleep(127) (~ the inverse transform that sits between the stores in the real kernel).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));

template <bool HALVES, bool NT>
__global__ void __launch_bounds__(1024) k_tiles(float2 *w4, long fstride, int P, int ntiles, int delay)
{
    const int bt = blockIdx.x;
    int tile = (bt & 7) * (gridDim.x >> 3) + (bt >> 3);            // adjacent tiles on one XCD, as k_col_full
    if (tile >= ntiles) return;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, l = lane >> 2, c = lane & 3;
    const float val = 1.0f + tid * 1e-6f;
    for (int pass = 0; pass < 4; ++pass) {
        if (!HALVES) {
            float2 *dst = w4 + (size_t)pass * fstride + (size_t)tile * 8 + 2 * c;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                f4v v = {val, val + i, val, val - i};
                f4v *p = reinterpret_cast<f4v *>(dst + (size_t)(256 * i + 16 * w + l) * P);
                if (NT) __builtin_nontemporal_store(v, p); else *p = v;
            }
        } else {
            const int f0 = 2 * (pass >> 1), set = pass & 1;
#pragma unroll
            for (int ff = 0; ff < 2; ++ff) {
                float2 *dst = w4 + (size_t)(f0 + ff) * fstride + (size_t)tile * 8 + c + 4 * set;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    f2v v = {val + ff, val + i};
                    f2v *p = reinterpret_cast<f2v *>(dst + (size_t)(256 * i + 16 * w + l) * P);
                    if (NT) __builtin_nontemporal_store(v, p); else *p = v;
                }
            }
        }
        for (int d = 0; d < delay; ++d) __builtin_amdgcn_s_sleep(127);
    }
}

template <typename F> static float timeit(F f, int reps)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); f(); hipDeviceSynchronize();
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main()
{
    const int NR = 4096, P = 2096, ntiles = 242;
    const size_t n = (size_t)NR * P;
    float2 *d; CK(hipMalloc(&d, n * sizeof(float2) * 8));          // two sets of four fields: alternate so that nothing stays cached
    CK(hipMemset(d, 0, n * sizeof(float2) * 8));
    const double bytes = 4.0 * NR * ntiles * 8 * 8;               // four fields x rows x 64 B per tile row
    int k = 0;
#define RUN(H, NT, DELAY) { float ms = timeit([&] { hipLaunchKernelGGL((k_tiles<H, NT>), dim3(256), dim3(1024), 0, 0, d + (size_t)(k++ & 1) * 4 * n, (long)n, P, ntiles, DELAY); }, 20); \
        printf("%s %s delay=%d : %.4f ms  %.0f GB/s\n", H ? "halves" : "full  ", NT ? "nt    " : "normal", DELAY, ms, bytes / ms / 1e6); }
    for (int delay : {0, 2, 4}) {
        RUN(false, true, delay) RUN(true, true, delay) RUN(false, false, delay) RUN(true, false, delay)
    }
    hipFree(d);
    return 0;
}
