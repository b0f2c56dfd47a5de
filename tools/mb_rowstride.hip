// Microbenchmark (design probe, not product): does k_col_strided<128, +1> at 16384^2 (4.8 TB/s against 5.9 TB/s for the same kernel
// at 4096^2) lose its bandwidth to the DISTANCE between the 128 rows of a wave tile (8 MB apart: 128 different 2 MB pages per tile,
// 4.3 GB under one launch)?  Same access shape as the kernel (16 columns x 128 rows per wave, 16-byte loads of 8 rows x 8 column pairs,
// 8-byte stores of 4 rows x 16 columns, in place, four fields, tiles of neighbouring columns on neighbouring waves), with the tile's
// rows S rows apart: row(c) = hi * 128 * S + c * S + lo, (hi, lo) = the other 128 values; S = 128 is the kernel's layout, S = 1 a
// contiguous block of 128 rows.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));

// OUT: 0 = in place (the kernel's way), otherwise the tile is written to the same position of a second set of arrays `OUT` bytes further on
__global__ void __launch_bounds__(256) k_tiles(float2 *data, long fstride, int P, int nct, int S, long ntiles, size_t out_off)
{
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = lane >> 3, cp = lane & 7, h = lane >> 4, c16 = lane & 15;
    for (long tile = (long)blockIdx.x * 4 + wv; tile < ntiles; tile += (long)gridDim.x * 4) {
        const int f = (int)(tile / (128L * nct));
        const int rem = (int)(tile - (long)f * 128 * nct);
        const int b = rem / nct, ct = rem - b * nct;
        const int lo = b % S, hi = b / S;
        float2 *base = data + (size_t)f * fstride + (size_t)(hi * 128 * S + lo) * P + ct * 16;
        f4v in[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) in[m] = *reinterpret_cast<const f4v *>(base + (size_t)(g + 8 * m) * S * P + 2 * cp);
        f4v acc = in[0];
#pragma unroll
        for (int m = 1; m < 16; ++m) acc += in[m];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int k = h + 4 * s + 16 * q;
                f2v o = {acc.x + (float)k, acc.y};
                *reinterpret_cast<f2v *>(base + out_off + (size_t)k * S * P + c16) = o;
            }
    }
}

// the same bytes per wave as 32 columns x 64 rows (256-byte row segments): would wider column tiles stream faster?
__global__ void __launch_bounds__(256) k_tiles_wide(float2 *data, long fstride, int P, int nct2, long ntiles)
{
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = lane >> 4, cp = lane & 15, h = lane >> 5, c32 = lane & 31;
    for (long tile = (long)blockIdx.x * 4 + wv; tile < ntiles; tile += (long)gridDim.x * 4) {
        const int f = (int)(tile / (256L * nct2));
        const int rem = (int)(tile - (long)f * 256 * nct2);
        const int b = rem / nct2, ct = rem - b * nct2;                // b: 256 blocks of 64 contiguous rows
        float2 *base = data + (size_t)f * fstride + (size_t)(b * 64) * P + ct * 32;
        f4v in[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) in[m] = *reinterpret_cast<const f4v *>(base + (size_t)(g + 4 * m) * P + 2 * cp);
        f4v acc = in[0];
#pragma unroll
        for (int m = 1; m < 16; ++m) acc += in[m];
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            const int k = h + 2 * s;
            f2v o = {acc.x + (float)k, acc.y};
            *reinterpret_cast<f2v *>(base + (size_t)k * P + c32) = o;
        }
    }
}

int main(int argc, char **argv)
{
    const int nx = 16384, P = 8208, nct = 488;                      // active 16-column tiles at 16384^2
    const size_t n = (size_t)nx * P;
    float2 *d; CK(hipMalloc(&d, 8 * n * sizeof(float2)));
    CK(hipMemset(d, 0, 8 * n * sizeof(float2)));
    const long ntiles = 4L * 128 * nct;
    const double bytes = 2.0 * ntiles * 128 * 128;                  // read + write, 128 rows x 128 B per tile
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int grids[] = {2048, 8192};
    for (int grid : grids)
        for (int S : {128, 32, 16, 4, 1}) {
            for (size_t off : {(size_t)0, 4 * n}) {
            for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k_tiles, dim3(grid), dim3(256), 0, 0, d, (long)n, P, nct, S, ntiles, off);
            (void)hipDeviceSynchronize(); (void)hipEventRecord(e0, 0);
            for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k_tiles, dim3(grid), dim3(256), 0, 0, d, (long)n, P, nct, S, ntiles, off);
            (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
            printf("grid %5d  rows %3d apart %s : %.3f ms per launch, %.0f GB/s\n", grid, S, off ? "out of place" : "in place    ", ms, bytes / ms / 1e6);
            }
        }
    {
        const int nct2 = nct / 2;
        const long nt = 4L * 256 * nct2;
        for (int grid : grids) {
            for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k_tiles_wide, dim3(grid), dim3(256), 0, 0, d, (long)n, P, nct2, nt);
            (void)hipDeviceSynchronize(); (void)hipEventRecord(e0, 0);
            for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k_tiles_wide, dim3(grid), dim3(256), 0, 0, d, (long)n, P, nct2, nt);
            (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
            printf("grid %5d  32 columns x 64 contiguous rows (256-byte segments) : %.3f ms per launch, %.0f GB/s\n", grid, ms, 2.0 * nt * 64 * 256 / ms / 1e6);
        }
    }
    CK(hipDeviceSynchronize());
    return 0;
}
