// Microbenchmark (design probe, not product): what a contiguous 16-B/lane stream achieves on this
// GPU for the read:write mixes of the four stage kernels -- the practical ceiling under the 8 TB/s
// nominal HBM figure that bench.py's roofline uses.
//   hipcc -w --offload-arch=gfx950 -O3 -o tools/mb_stream tools/mb_stream.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// NRD input arrays are summed into NWR output arrays, chunk by chunk (UNR float4 in flight per array per lane)
template <int NRD, int NWR, int UNR, int NT>
__global__ void __launch_bounds__(256) k_mix(const f4 *__restrict__ in, f4 *__restrict__ out, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256 * UNR;
    for (size_t base = (size_t)blockIdx.x * 256 * UNR + threadIdx.x; base < n; base += stride) {
        f4 v[NRD > 0 ? NRD : 1][UNR];
#pragma unroll
        for (int a = 0; a < NRD; ++a)
#pragma unroll
            for (int u = 0; u < UNR; ++u) v[a][u] = (NT & 1) ? __builtin_nontemporal_load(&in[(size_t)a * n + base + u * 256]) : in[(size_t)a * n + base + u * 256];
        f4 s[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            s[u] = (f4){1.f, 2.f, 3.f, 4.f};
#pragma unroll
            for (int a = 0; a < NRD; ++a) s[u] += v[a][u];
        }
        if (NWR == 0) {
            float t = 0.f;
#pragma unroll
            for (int u = 0; u < UNR; ++u) t += s[u].x + s[u].y + s[u].z + s[u].w;
            if (t == 123.456f) out[0] = s[0];
        }
#pragma unroll
        for (int a = 0; a < NWR; ++a)
#pragma unroll
            for (int u = 0; u < UNR; ++u) { if (NT & 2) __builtin_nontemporal_store(s[u], &out[(size_t)a * n + base + u * 256]); else out[(size_t)a * n + base + u * 256] = s[u]; }
    }
}

template <int NRD, int NWR, int UNR, int NT = 0>
static int run(const f4 *in, f4 *out, size_t n, int blocks)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k_mix<NRD, NWR, UNR, NT><<<blocks, 256>>>(in, out, n); CK(hipDeviceSynchronize());
    const int reps = 10;
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) k_mix<NRD, NWR, UNR, NT><<<blocks, 256>>>(in, out, n);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= reps;
    const double bytes = (double)(NRD + NWR) * n * 16;
    printf("nt=%d read %d : write %d arrays of %4zu MiB, unroll %d, %5d blocks: %7.3f ms  %6.0f GB/s\n", NT, NRD, NWR, n * 16 >> 20, UNR, blocks, ms, bytes / ms * 1e-6);
    return 0;
}

int main()
{
    const size_t nmax = (size_t)16 << 20;              // 256 MiB per array = 16 Mi float4
    f4 *in, *out;
    CK(hipMalloc(&in, nmax * 16 * 4)); CK(hipMalloc(&out, nmax * 16 * 5));
    CK(hipMemset(in, 0x3c, nmax * 16 * 4)); CK(hipMemset(out, 0, nmax * 16 * 5));
    const int blocks = 8192;
    for (size_t n : {nmax, nmax / 4}) {     // 256, 64, 16 MiB per array: beyond / around / inside the 256 MiB MALL
        run<1, 0, 4>(in, out, n, blocks);   // read only
        run<4, 0, 2>(in, out, n, blocks);
        run<0, 1, 4>(in, out, n, blocks);   // write only
        run<1, 1, 4>(in, out, n, blocks);   // copy (strided passes: 1:1)
        run<4, 1, 2>(in, out, n, blocks);   // row kernel: 4:1
        run<3, 5, 2>(in, out, n, blocks);   // middle kernel: ~3:5
        run<1, 1, 4, 1>(in, out, n, blocks); run<1, 1, 4, 2>(in, out, n, blocks); run<1, 1, 4, 3>(in, out, n, blocks);
        run<4, 1, 2, 3>(in, out, n, blocks); run<3, 5, 2, 3>(in, out, n, blocks); run<3, 5, 2, 2>(in, out, n, blocks);
    }
    return 0;
}
