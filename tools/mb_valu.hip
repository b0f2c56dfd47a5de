// Microbenchmark (design probe, not product): issue cost of packed-f32 vs scalar-f32 vector instructions on gfx950.
//   hipcc -w --offload-arch=gfx950 -O3 -o tools/mb_valu tools/mb_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND>
__global__ void __launch_bounds__(256) k(float *out, int iters)
{
    v2f a0 = {1.f + threadIdx.x, 2.f}, a1 = {3.f, 4.f}, a2 = {5.f, 6.f}, a3 = {7.f, 8.f}, b = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
    float s0 = a0.x, s1 = a1.x, s2 = a2.x, s3 = a3.x, s4 = a0.y, s5 = a1.y, s6 = a2.y, s7 = a3.y, sb = b.x, sc = c.x;
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {        // 8 independent scalar fma chains: 8 v_fma_f32 per group
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                              "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                              : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4), "+v"(s5), "+v"(s6), "+v"(s7) : "v"(sb), "v"(sc));)
        } else if (KIND == 1) { // 4 independent packed fma chains: 4 v_pk_fma_f32 per group (same flops as 8 scalar)
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)
        } else if (KIND == 2) { // packed add
            REP8(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));)
        } else if (KIND == 3) { // packed fma with op_sel swizzle (as in cmul)
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %5 op_sel_hi:[0,1,1] neg_lo:[0,0,1]\n v_pk_fma_f32 %1, %1, %4, %5 op_sel_hi:[0,1,1] neg_lo:[0,0,1]\n"
                              "v_pk_fma_f32 %2, %2, %4, %5 op_sel_hi:[0,1,1] neg_lo:[0,0,1]\n v_pk_fma_f32 %3, %3, %4, %5 op_sel_hi:[0,1,1] neg_lo:[0,0,1]\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)
        } else if (KIND == 4) { // scalar add
            REP8(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                              "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                              : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4), "+v"(s5), "+v"(s6), "+v"(s7) : "v"(sc));)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7 + a0.x + a0.y + a1.x + a1.y + a2.x + a2.y + a3.x + a3.y;
}

template <int KIND> static void run(const char *name, float *out, int wgs_per_cu, int ninstr_per_iter, int lanes_ops_per_instr)
{
    const int iters = 2000, blocks = 256 * wgs_per_cu;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<KIND><<<blocks, 256>>>(out, 10); hipDeviceSynchronize();
    hipEventRecord(a, 0);
    k<KIND><<<blocks, 256>>>(out, iters);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // per SIMD: wgs_per_cu waves (256-thread WG = 4 waves = 1 per SIMD); each executes iters*ninstr instructions
    const double instr_per_simd = (double)wgs_per_cu * iters * ninstr_per_iter;
    printf("%-26s %d waves/SIMD: %7.3f ms  %5.2f ns per instr per SIMD  (%5.2f cycles at 2.4 GHz)  %6.1f Tflop-lane-ops/s\n", name, wgs_per_cu, ms,
           ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4, instr_per_simd * 1024 * 64 * lanes_ops_per_instr / (ms * 1e-3) * 1e-12);
}

int main()
{
    float *out; hipMalloc(&out, 256 * 8 * 256 * 4);
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_fma_f32", out, w, 64, 1);
        run<4>("v_add_f32", out, w, 64, 1);
        run<1>("v_pk_fma_f32", out, w, 32, 2);
        run<3>("v_pk_fma_f32 op_sel/neg", out, w, 32, 2);
        run<2>("v_pk_add_f32", out, w, 32, 2);
    }
    return 0;
}
