// Microbenchmark (design probe, not product): HBM efficiency of column-group streaming with
// 32/64/128-byte row segments on a [4096][2064] complex64 array -- decides whether a single-pass
// x-transform (whole columns resident in registers) is viable on MI355X.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// one workgroup = one group of B columns, all NR rows; thread holds NR*B/THREADS complex
template <int B, int THREADS, int NR, bool WRITE>
__global__ void __launch_bounds__(THREADS) k_group(float2 *data, int P, int ngroups, int xcd_aware)
{
    constexpr int LPR = B / 2;                 // lanes per row (16 B per lane)
    constexpr int RPI = THREADS / LPR;         // rows per instruction across the workgroup
    constexpr int NI = NR / RPI;               // float4 per thread
    int g = blockIdx.x;
    if (xcd_aware) { int per = (ngroups + 7) / 8; g = (blockIdx.x % 8) * per + blockIdx.x / 8; }
    if (g >= ngroups) return;
    const int r0 = threadIdx.x / LPR, c = threadIdx.x % LPR;
    float4 v[NI];
    float2 *base = data + (size_t)g * B + 2 * c;
#pragma unroll
    for (int i = 0; i < NI; ++i) v[i] = *reinterpret_cast<const float4 *>(base + (size_t)(r0 + i * RPI) * P);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) s += v[i].x + v[i].y + v[i].z + v[i].w;
    if (WRITE) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            float4 o = v[i]; o.x += s * 1e-30f;
            *reinterpret_cast<float4 *>(base + (size_t)(r0 + i * RPI) * P) = o;
        }
    } else if (s == 123.456f) data[0].x = s;
}

// reference: fully contiguous copy-in-place
__global__ void __launch_bounds__(256) k_stream(float4 *d, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = d[i]; v.x += 1e-30f; d[i] = v;
    }
}

template <typename F> static float timeit(F f, int reps)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main()
{
    const int NR = 4096, P = 2064;
    const size_t n = (size_t)NR * P;
    float2 *d; CK(hipMalloc(&d, n * sizeof(float2) * 4));   // 4 fields so that we exceed the 256 MiB MALL
    CK(hipMemset(d, 0, n * sizeof(float2) * 4));
    const double bytes1 = (double)n * 8;
    float ms;
    ms = timeit([&] { hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, 0, (float4 *)d, n * 4 / 2); }, 10);
    printf("stream r+w 4 fields        : %.3f ms  %.0f GB/s\n", ms, 8 * bytes1 / ms / 1e6);
#define RUN(B, T, W, X)                                                                                     \
    ms = timeit([&] { for (int f = 0; f < 4; ++f) hipLaunchKernelGGL((k_group<B, T, NR, W>), dim3((P / B + 7) / 8 * 8), dim3(T), 0, 0, d + f * n, P, P / B, X); }, 10); \
    printf("B=%2d (%3d B seg) T=%4d %s xcd=%d : %.3f ms/4 fields  %.0f GB/s\n", B, B * 8, T, W ? "r+w" : "r  ", X, ms, (W ? 8 : 4) * bytes1 / ms / 1e6);
    RUN(8, 1024, false, 0) RUN(8, 1024, false, 1) RUN(8, 1024, true, 0) RUN(8, 1024, true, 1)
    RUN(4, 1024, false, 0) RUN(4, 1024, false, 1) RUN(4, 1024, true, 0) RUN(4, 1024, true, 1)
    RUN(4, 512, false, 0) RUN(4, 512, true, 1)
    RUN(16, 1024, false, 0) RUN(16, 1024, true, 0)
    hipFree(d);
    return 0;
}
