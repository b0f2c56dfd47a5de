// fb_kernels.h -- HIP kernels of the RK4 hot path (gfx950).
//
// Data layouts in HBM (all private to the engine; the C ABI converts at its boundary):
//   "mixed"    arrays  T, W4[f]   : [row = x][col = ky], pitch P complex (P multiple of 16, >= ny/2+1,
//                                   pad columns are zero and stay zero: every pass is linear)
//   "spectral" arrays  ZA, ZB, ACC: [row = N2*c + d][col = ky], holding kx = c + N1*d  (nx = N1*N2)
// The x-direction (column) transform of length nx is done as two sub-passes of lengths N1 and
// N2 on wave tiles of 16 columns, so that every HBM access is a full 128-byte line:
//   forward : k_col_strided<N1,-1> (over a; rows N2*a+b)  -> twiddle W_nx^{bc} -> block FFT over b
//   backward: block FFT over d -> twiddle conj -> k_col_strided<N1,+1> (over c)
// The block sub-passes of both directions, the viscous term, the dealiasing mask, the RK4 update
// and the four spectral derivatives are fused in k_col_mid.
// The y-direction (row) transforms, normalisation, sign of u and the Jacobian product are fused
// in k_row<FUSED> (a7-a11).
#pragma once
#include "fb_fft_core.h"

// -------------------------------------------------------------------------------------------
// spectral coefficient tables (device): built once per context from fftwfop.cpp:15-24 values
// -------------------------------------------------------------------------------------------
struct SpecCoef {
    const float  *gx;      // [nx]  gradx_coe                       fftwfop.cpp:15-20
    const double *kx2;     // [nx]  (double)gradx_coe^2             fftwfop.cpp:42,45 (pow(float,int))
    const float  *gy;      // [P]   grady_coe, zero in pad columns  fftwfop.cpp:22-24
    const double *ky2;     // [P]
    double gws;            // generalized_wavenumber_square         fftwfop.cpp:57
    int nx, hy;            // hy = ny/2+1 (columns >= hy are padding)
    int gws_i;             // ceil(gws): ii^2 + j^2 is an integer below 2^28, so  (double) r2 >= gws  <=>  r2 >= ceil(gws)
};

// laplacian_coe[i][j] = (float)-(kx2 + ky2)                         fftwfop.cpp:45
FB_DEV float coef_lap(const SpecCoef &c, int i, int j) { return (float)(-(c.kx2[i] + c.ky2[j])); }
// dealiasing_mask                                                    fftwfop.cpp:57-68
FB_DEV float coef_mask(const SpecCoef &c, int i, int j)
{
    const int ii = i < c.nx - i ? i : c.nx - i;
    const int r2 = ii * ii + j * j;                    // exact: ii <= nx/2 <= 8192, j <= 8200 (the reference compares the same integers as doubles)
    return (r2 >= c.gws_i || j >= c.hy) ? 0.0f : 1.0f;
}

// -------------------------------------------------------------------------------------------
// pointwise kernels of the standalone operator API (bit-exact float32 forms; no contraction)
// -------------------------------------------------------------------------------------------
enum { OP_GRADX = 0, OP_GRADY = 1, OP_LAP = 2, OP_INVLAP = 3, OP_DEALIAS = 4 };

template <int OP>
__global__ void __launch_bounds__(256) k_spec_op(SpecCoef c, const cf *__restrict__ in, cf *__restrict__ out, int hy, size_t total)
{
#pragma clang fp contract(off)
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(idx / hy), j = (int)(idx - (size_t)i * hy);
        cf a = in[idx], r;
        if (OP == OP_GRADX) { float k = c.gx[i]; r = cf_make(-a.y * k, a.x * k); }            // fftwfop.cpp:87-94
        else if (OP == OP_GRADY) { float k = c.gy[j]; r = cf_make(-a.y * k, a.x * k); }       // :96-103
        else if (OP == OP_LAP) { float k = coef_lap(c, i, j); r = cf_make(a.x * k, a.y * k); } // :105-110
        else if (OP == OP_INVLAP) { float k = (i == 0 && j == 0) ? 1.0f : coef_lap(c, i, j);   // :112-117
                                    r = cf_make(a.x / k, a.y / k); }
        else { float k = coef_mask(c, i, j); r = cf_make(a.x * k, a.y * k); }                 // :119-124
        out[idx] = r;
    }
}

__global__ void __launch_bounds__(256) k_scale_real(float *d, float s, int divide, size_t n)
{
#pragma clang fp contract(off)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        d[i] = divide ? d[i] / s : d[i] * s;
}

// main.cpp:225-227   dvortdt = - u*dvortdx - v*dvortdy + vort_src
__global__ void __launch_bounds__(256) k_jacobian(const float *__restrict__ u, const float *__restrict__ v,
                                                  const float *__restrict__ dx, const float *__restrict__ dy,
                                                  const float *__restrict__ src, float *__restrict__ out, size_t n)
{
#pragma clang fp contract(off)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float s = src ? src[i] : 0.0f;
        out[i] = -u[i] * dx[i] - v[i] * dy[i] + s;
    }
}

// mode 0: acc += x*a (main.cpp:240-243) ; mode 1: out = base + x*a (main.cpp:246-251)
__global__ void __launch_bounds__(256) k_spec_axpy(const float *__restrict__ base, const float *__restrict__ x, float a,
                                                   float *__restrict__ out, size_t n)
{
#pragma clang fp contract(off)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = base[i] + x[i] * a;
}

// main.cpp:309-312
__global__ void __launch_bounds__(256) k_rk4_combine(const float *__restrict__ base, const float *__restrict__ k1,
                                                     const float *__restrict__ k2, const float *__restrict__ k3,
                                                     const float *__restrict__ k4, float dt, float *__restrict__ out, size_t n)
{
#pragma clang fp contract(off)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = base[i] + (k1[i] + 2.0f * k2[i] + 2.0f * k3[i] + k4[i]) * dt / 6.0f;
}

// natural half spectrum [kx][ky] pitch hy  <->  private spectral layout [N2*c+d][ky] pitch P
template <bool TO_PRIVATE>
__global__ void __launch_bounds__(256) k_spec_relayout(const cf *__restrict__ in, cf *__restrict__ out, int nx, int hy, int P, int N1, int N2)
{
    const size_t total = (size_t)nx * P;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(idx / P), col = (int)(idx - (size_t)row * P);
        const int c = row / N2, d = row - c * N2, kx = c + N1 * d;
        if (TO_PRIVATE) out[idx] = col < hy ? in[(size_t)kx * hy + col] : cf_make(0.f, 0.f);
        else if (col < hy) out[(size_t)kx * hy + col] = in[idx];
    }
}

// record path of the multi-GPU model: psi_c = invertLaplacian(vort_c) and its derivatives, in place on one column group's
// 3-pass private layout (row N2*c + d holds kx = c + N1*d; local column j holds ky = ky0 + j).  MODE 0: psi_c (main.cpp:179),
// 1: grady(psi_c) (main.cpp:198), 2: gradx(psi_c) (main.cpp:212).  Same float32 forms as k_spec_op (no contraction).
template <int MODE>
__global__ void __launch_bounds__(256) k_psi_private(SpecCoef c, cf *__restrict__ z, int P, int N1, int N2, int ky0)
{
#pragma clang fp contract(off)
    const size_t total = (size_t)c.nx * P;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(idx / P), col = (int)(idx - (size_t)row * P);
        const int cc = row / N2, d = row - cc * N2, i = cc + N1 * d, j = ky0 + col;
        cf a = z[idx];
        if (j >= c.hy) { z[idx] = cf_make(0.f, 0.f); continue; }                      // pad columns
        const float li = (i == 0 && j == 0) ? 1.0f : coef_lap(c, i, j);               // fftwfop.cpp:42-43,112-117
        a = cf_make(a.x / li, a.y / li);
        if (MODE == 1) { const float k = c.gy[j]; a = cf_make(-a.y * k, a.x * k); }   // fftwfop.cpp:96-103
        if (MODE == 2) { const float k = c.gx[i]; a = cf_make(-a.y * k, a.x * k); }   // fftwfop.cpp:87-94
        z[idx] = a;
    }
}

// State arrays (vort_c0, stage state, RK accumulator) are touched by k_col_mid only, so they live
// in that kernel's register order ("tile-major"): tile (cb, ct) = N2 rows x 16 columns is contiguous,
//   complex index = ((tile*(NLB/2) + e/2)*64 + lane)*2 + (e & 1),   e = 8 s + q  <->  row d = h + 4 s + R1 q,
// lane = 16 h + c.  One float4 per lane then carries two elements and every wave access is 1 KiB
// contiguous.  Used when R1 = N2/8 >= 4; smaller blocks keep the row layout (all lanes would not be active).
template <bool TO_TM>
__global__ void __launch_bounds__(256) k_state_relayout(const cf *__restrict__ in, cf *__restrict__ out, int nx, int P, int N2)
{
    const int R1 = N2 >> 3, NLB = N2 >> 2, ntc = P >> 4;
    const size_t total = (size_t)nx * P;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(idx / P), col = (int)(idx - (size_t)row * P);
        const int cb = row / N2, d = row - cb * N2, ct = col >> 4, c = col & 15;
        const int q = d / R1, rem = d - q * R1, h = rem & 3, sidx = rem >> 2, e = 8 * sidx + q;
        const size_t tile = (size_t)cb * ntc + ct;
        const size_t tm = ((tile * (NLB / 2) + (e >> 1)) * 64 + (16 * h + c)) * 2 + (e & 1);
        if (TO_TM) out[tm] = in[idx]; else out[idx] = in[tm];
    }
}

// -------------------------------------------------------------------------------------------
// row pass (y direction), T = N/16 threads per transform, G = max(1, 256/T) row pairs per WG
// -------------------------------------------------------------------------------------------
enum { ROW_FUSED = 0, ROW_INV = 1, ROW_FWD = 2 };

// A mixed-space array as the row pass sees it: element (field, row, k), k = global ky in [0, ny/2].
// One GPU: one segment of pitch ka.  Multi-GPU exchange buffers: the row is cut into ky slabs of `ka` columns (the ACTIVE
// columns, ky < katot = world*ka, exchanged every RK stage) followed by slabs of `kf` columns (the FROZEN columns beyond the
// dealiasing circle, SURVEY note N1, exchanged once).  Where the stage is pipelined by column groups (fb_slab_driver.h) an
// active slab is held in two buffers: its first ka0 columns in segment a (pitch ka0), the other kb = ka - ka0 in segment b
// (pitch kb); kb == 0: one buffer, ka0 == ka.  Slab s of a segment lives sstr* complex after slab 0.  Divisions by ka / kf
// are multiplications by mag* = floor(2^32/d) + 1 (exact for k*d < 2^32).
struct RowView {
    const cf *a, *b, *f;        // segment bases (field 0): active columns [0, ka0) of a slab / [ka0, ka) / frozen
    long fstrA, fstrB, fstrF;   // field strides (complex)
    long sstrA, sstrB, sstrF;   // slab strides (complex); unused on one GPU
    int ka, ka0, kb, kf;        // one GPU: ka = ka0 = pitch
    int katot;
    unsigned magA, magF;
};
struct RowArgs {
    RowView M;              // mixed-space inputs                                   (FUSED: 4 fields, INV: 1)
    RowView T;              // mixed-space output                                   (FUSED, FWD)
    int t_frozen;           // multi-GPU: 1 = also store the frozen columns of T (FWD: the state); FUSED: 0, their tendency is masked
    const float *src;       // vort_src (real [x][y]) or NULL                       (FUSED)
    const int *src_nz;      // per local row: 1 = the row of vort_src holds a non-zero value; rows of zeros are not read (x + 0 == x)
    const float *rin;       // real input  [x][y]                                   (FWD)
    float *rout;            // real output [x][y]                                   (INV)
    int x0, nx;             // local rows [x0, x0 + nx), nx even (a row chunk of the pipelined multi-GPU step, else everything)
    float scale;            // 1/GRIDS (FUSED) ; 1/GRIDS or 1 (INV)
    int prescaled;          // FUSED, k_rowq on one GPU: the four fields arrive multiplied by 1/GRIDS already (FullArgs::wscale)
    const cf *tw_bwd, *tw_fwd;
    // k_rowh2 (nx = 8192 with the radix-2 x step fused into the row pass, fb_col_full.h): rows per sub-sequence and W_nx^j
    long sub_rows;
    const cf *tw_x;
};

// one workgroup per row of vort_src: does the row hold anything but zeros?  (The FIFO producer's cake covers a tenth of the rows,
// and its "switch off" input is a field of zeros, vort_src_input.cpp:46,52-55: such rows cost the row pass no traffic.)
__global__ void __launch_bounds__(256) k_src_row_flags(const float *__restrict__ src, int *__restrict__ flags, int ny)
{
    const float4 *row = reinterpret_cast<const float4 *>(src + (size_t)blockIdx.x * ny);
    int any = 0;
    for (int i = threadIdx.x; i < ny / 4; i += 256) { const float4 v = row[i]; any |= (v.x != 0.f) | (v.y != 0.f) | (v.z != 0.f) | (v.w != 0.f); }
    any = __syncthreads_or(any);
    if (threadIdx.x == 0) flags[blockIdx.x] = any ? 1 : 0;
}

template <int N> struct RowCfg {
    static constexpr int T = N / 16;
    static constexpr int G = T >= 256 ? 1 : 256 / T;
    static constexpr int THREADS = T * G;
    // half-buffer exchange (row_fft_half): 51 KiB per workgroup at N = 4096 -> three workgroups per CU
    static constexpr bool HALFX = false;   // measured at N = 4096: 0.205 ms vs 0.104 ms with the full buffer
    static constexpr int LSTR = HALFX ? (N / 2 + N / 32) : (N + N / 16);   // padded complex per group (exchange buffer)
    // LDS-DMA prefetch of the next phase's two half-spectrum rows (fused mode): groups must be
    // whole waves and buffer + staging must leave room for two workgroups per CU
    static constexpr bool DMA = N >= 1024 && N <= 8192;
    static constexpr int GSTR = LSTR + (DMA ? N : 0);     // complex per group incl. staging [A: N/2][B: N/2]
    static constexpr bool RES = RowRes<N>::value;
    static constexpr bool SHARE = RES && RowPlanSymmetric<N>::value;        // one twiddle set for both directions
    static constexpr int TWL_B = RowTwSrc<N, false, RES>::LDS_CF, TWL_F = SHARE ? 0 : RowTwSrc<N, true, RES>::LDS_CF;
    // Small grids (N <= 1024: a few hundred row pairs for 256 CUs) are bound by the chain of five dependent transforms a row pair
    // takes in one thread group.  There the fused pass gives a pair TWO groups, one per x row, side by side: each runs its row's two
    // backward transforms, they hand their tendency values to each other through LDS, both run the packed forward transform (its
    // barriers must be met by every wave), the even group stores.  (As k_row3 at 768^2: 0.054 -> 0.035 ms per launch.)
#ifndef FB_PAIR2_MAXN
#define FB_PAIR2_MAXN 1024   /* 2048^2 has enough row pairs to fill the chip: measured below */
#endif
    static constexpr bool PAIR2 = N <= FB_PAIR2_MAXN && G >= 2;
    static constexpr int VALB = PAIR2 ? G * 8 * T : 0;                     // complex: 16 floats per thread and group
    static constexpr size_t LDS_BYTES = ((size_t)G * GSTR + TWL_B + TWL_F + VALB) * sizeof(cf);
    static constexpr int MIN_WAVES = HALFX ? 3 : (THREADS == 256 ? 2 : (THREADS == 512 ? 2 : 4));
};

// element (field, row, k) of a mixed-space array (see RowView); SLAB = multi-GPU exchange buffer
template <bool SLAB>
FB_DEV const cf *row_ptr(const RowView &v, int field, int row, int k)
{
    if (!SLAB) return v.a + (size_t)field * v.fstrA + (size_t)row * v.ka + k;
    if (k < v.katot) {
        const int s = (int)__umulhi((unsigned)k, v.magA), w = k - s * v.ka;          // slab (peer rank), column within the slab
        if (v.kb == 0) return v.a + (size_t)field * v.fstrA + (size_t)s * v.sstrA + (size_t)row * v.ka + w;      // (uniform) one column group
        const bool inb = w >= v.ka0;
        const cf *base = inb ? v.b : v.a;
        const long fstr = inb ? v.fstrB : v.fstrA, sstr = inb ? v.sstrB : v.sstrA;
        const int kc = inb ? v.kb : v.ka0, wc = inb ? w - v.ka0 : w;
        return base + (size_t)field * fstr + (size_t)s * sstr + (size_t)row * kc + wc;
    }
    const int kk = k - v.katot, s = (int)__umulhi((unsigned)kk, v.magF);
    return v.f + (size_t)field * v.fstrF + (size_t)s * v.sstrF + (size_t)row * v.kf + (kk - s * v.kf);
}
// does column k of an output view get stored?  (multi-GPU fused pass: the frozen columns' tendency is never used)
template <bool SLAB> FB_DEV bool row_keep(const RowView &v, int t_frozen, int k) { return !SLAB || t_frozen || k < v.katot; }

// Hermitian-extend two half-spectrum rows A,B into Z = A_ext + i B_ext, straight into the first
// backward stage's registers (SURVEY note N2: imaginary parts at k=0 and k=N/2 are ignored).
// Thread t owns positions t + i*T: for i < 8 that is k itself, for i >= 8 the mirror of N - pos.
template <int N, bool SLAB>
FB_DEV void row_load_pair(cf *reg, int t, const RowView &v, int fA, int fB, int rowA, int rowB)
{
    constexpr int T = N / 16, R0 = RowTw<N, false>::radix(0);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int i = ord_i<R0>(e);
        if (i < 8) {
            const int k = t + i * T;
            const cf a = *row_ptr<SLAB>(v, fA, rowA, k), b = *row_ptr<SLAB>(v, fB, rowB, k);
            reg[e] = (i == 0 && t == 0) ? cf_make(a.x, b.x) : cf_make(a.x - b.y, a.y + b.x);
        } else {
            const int k = (16 - i) * T - t;                 // in (0, N/2]; == N/2 only for t == 0, i == 8
            const cf a = *row_ptr<SLAB>(v, fA, rowA, k), b = *row_ptr<SLAB>(v, fB, rowB, k);
            reg[e] = (i == 8 && t == 0) ? cf_make(a.x, b.x) : cf_make(a.x + b.y, b.x - a.y);
        }
    }
}

// untangle Z = FFT(t0 + i t1), held in the last forward stage's register order, into the two
// half spectra and store rows rowA,rowB of T.  Only the upper half (positions >= N/2) goes
// through LDS: the mirror of k = t + i*T (i < 8) is position (16-i)*T - t.
template <int N, bool SLAB>
FB_DEV void row_store_pair(cf *lds, int t, const cf *reg, bool valid, const RowView &v, int t_frozen, int rowA, int rowB)
{
    constexpr int T = N / 16, RL = RowTw<N, true>::radix(RowPlan<N>::S - 1);
    constexpr int HOFF = RowCfg<N>::HALFX ? N / 2 : 0;           // only positions >= N/2 go through LDS
    lds_barrier();
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int i = ord_i<RL>(e);
        if (i >= 8) lds[lds_pad(t + i * T - HOFF)] = reg[e];
    }
    lds_barrier();
    if (!valid) return;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int i = ord_i<RL>(e);
        if (i < 8) {
            const int k = t + i * T;
            const cf zk = reg[e];
            const cf zn = (i == 0 && t == 0) ? zk : lds_rd(&lds[lds_pad(N - k - HOFF)]);
            if (!row_keep<SLAB>(v, t_frozen, k)) continue;
            st2<(FB_NT & 8) != 0>(const_cast<cf *>(row_ptr<SLAB>(v, 0, rowA, k)), cf_make(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y)));
            st2<(FB_NT & 8) != 0>(const_cast<cf *>(row_ptr<SLAB>(v, 0, rowB, k)), cf_make(0.5f * (zk.y + zn.y), 0.5f * (zn.x - zk.x)));
        } else if (i == 8 && t == 0 && row_keep<SLAB>(v, t_frozen, N / 2)) {      // Nyquist: its own mirror
            *const_cast<cf *>(row_ptr<SLAB>(v, 0, rowA, N / 2)) = cf_make(reg[e].x, 0.f);
            *const_cast<cf *>(row_ptr<SLAB>(v, 0, rowB, N / 2)) = cf_make(reg[e].y, 0.f);
        }
    }
}

// ---- LDS-DMA prefetch of two half-spectrum rows into the group's staging area ----------------
// stg[0..N/2) = row A (k = 0..N/2-1), stg[N/2..N) = row B; the Nyquist elements travel in registers.
template <int N, bool SLAB>
FB_DEV void row_dma_issue(cf *stg, int t, const RowView &v, int fA, int fB, int rowA, int rowB, cf &nyqA, cf &nyqB)
{
    constexpr int T = N / 16, NW = T >= 64 ? T / 64 : 1, CH = N / 256;   // waves per group, 1-KiB chunks per row (only used when T >= 64)
    const int w = t >> 6, lane = t & 63;
#pragma unroll
    for (int c = 0; c < CH / NW; ++c) {
        const int ch = w + c * NW, k = ch * 128 + lane * 2;
        cf *dstA = stg + ch * 128, *dstB = stg + N / 2 + ch * 128;   // wave-uniform; the DMA adds lane*16 B
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)row_ptr<SLAB>(v, fA, rowA, k),
                                         (void __attribute__((address_space(3))) *)dstA, 16, 0, (FB_NT & 4) ? 2 : 0);
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)row_ptr<SLAB>(v, fB, rowB, k),
                                         (void __attribute__((address_space(3))) *)dstB, 16, 0, (FB_NT & 4) ? 2 : 0);
    }
    if (t == 0) { nyqA = *row_ptr<SLAB>(v, fA, rowA, N / 2); nyqB = *row_ptr<SLAB>(v, fB, rowB, N / 2); }
}

// Hermitian extension from the staging area into the first backward stage's registers
template <int N>
FB_DEV void row_ext_from_stage(cf *reg, int t, const cf *stg, cf nyqA, cf nyqB)
{
    constexpr int T = N / 16, R0 = RowTw<N, false>::radix(0);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int i = ord_i<R0>(e);
        if (i < 8) {
            const int k = t + i * T;
            const cf a = lds_rd(&stg[k]), b = lds_rd(&stg[N / 2 + k]);
            reg[e] = (i == 0 && t == 0) ? cf_make(a.x, b.x) : cf_make(a.x - b.y, a.y + b.x);
        } else {
            int k = (16 - i) * T - t;
            const bool nyq = (i == 8 && t == 0);
            if (nyq) k = 0;
            cf a = lds_rd(&stg[k]), b = lds_rd(&stg[N / 2 + k]);
            reg[e] = nyq ? cf_make(nyqA.x, nyqB.x) : cf_make(a.x + b.y, b.x - a.y);
        }
    }
}

template <int N, bool FWD, class SRC>
FB_DEV void rowfft(cf *lds, int t, const SRC &src, cf *reg)
{
    if constexpr (RowCfg<N>::HALFX) row_fft_half<N, FWD>(lds, t, src, reg);
    else row_fft<N, FWD>(lds, t, src, reg);
}

template <int N, int MODE, bool SLAB>
__global__ void __launch_bounds__(RowCfg<N>::THREADS, RowCfg<N>::MIN_WAVES) k_row(RowArgs a)
{
    using C = RowCfg<N>;
    constexpr int T = C::T, G = C::G;
    constexpr int RL = RowTw<N, false>::radix(RowPlan<N>::S - 1);      // physical-space register order
    constexpr bool DMA = C::DMA && MODE == ROW_FUSED;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf *smem = reinterpret_cast<cf *>(smem_raw);
    const int grp = threadIdx.x / T, t = threadIdx.x - grp * T;
    cf *lds = smem + (size_t)grp * C::GSTR;
    cf *stg = lds + C::LSTR;
    const int npairs = a.nx >> 1;
    constexpr bool PAIR2 = C::PAIR2 && MODE == ROW_FUSED;      // two groups per row pair, one per x row (RowCfg)
    constexpr int PPW = PAIR2 ? G / 2 : G;                      // row pairs per workgroup
    const int pgrp = PAIR2 ? grp >> 1 : grp, half = PAIR2 ? grp & 1 : 0;
    const int iters = (npairs + gridDim.x * PPW - 1) / (gridDim.x * PPW);

    // stage twiddles: once per workgroup (registers + a small LDS table) for N <= 4096, streamed otherwise
    constexpr bool SHARE = C::SHARE;
    cf *twl = smem + (size_t)G * C::GSTR;
    float *valb = reinterpret_cast<float *>(twl + C::TWL_B + C::TWL_F);
    RowTwSrc<N, false, C::RES> twb;
    twb.init(a.tw_bwd, twl, t, threadIdx.x, C::THREADS);
    RowTwSrc<N, true, C::RES> twf_own;
    if (!SHARE) twf_own.init(a.tw_fwd, twl + C::TWL_B, t, threadIdx.x, C::THREADS);
    __syncthreads();

#ifdef FB_ROW_SAMEROW   /* timing experiment only: every workgroup works on rows 0,1 (no HBM traffic); results are wrong */
    auto pair_of = [&](int it, bool &valid) { const int pr = (it * gridDim.x + blockIdx.x) * PPW + pgrp; valid = pr < npairs; return a.x0; };
#else
    auto pair_of = [&](int it, bool &valid) { const int pr = (it * gridDim.x + blockIdx.x) * PPW + pgrp; valid = pr < npairs; return a.x0 + (valid ? 2 * pr : 0); };
#endif
    cf nyqA = cf_make(0.f, 0.f), nyqB = nyqA;
    if (DMA && iters > 0) {                                   // prologue: phase 0 of the first pair
        bool v; const int x = pair_of(0, v) + half;
        row_dma_issue<N, SLAB>(stg, t, a.M, 0, 1, x, x, nyqA, nyqB);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

    for (int it = 0; it < iters; ++it) {
        bool valid;
        const int x0 = pair_of(it, valid), x1 = x0 + 1;       // invalid groups recompute pair 0, store nothing
        cf reg[16];
        const int t_it = launder(t);   (void)t_it;

        if constexpr (PAIR2) {
            // this group's x row of the pair (x0 + half); the other group takes the other row at the same time
            const int x = x0 + half;
            float zx[16], zy[16];
            if (DMA) {                                            // (this row's first two fields were waited for before the previous stores)
                lds_barrier();
                row_ext_from_stage<N>(reg, launder(t), stg, nyqA, nyqB);
                lds_barrier();
                row_dma_issue<N, SLAB>(stg, launder(t), a.M, 2, 3, x, x, nyqA, nyqB);
            } else row_load_pair<N, SLAB>(reg, launder(t), a.M, 0, 1, x, x);
            rowfft<N, false>(lds, launder(t), twb, reg);
#pragma unroll
            for (int e = 0; e < 16; ++e) { zx[e] = reg[e].x * a.scale; zy[e] = reg[e].y * a.scale; }   // main.cpp:154,168
            if (DMA) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                lds_barrier();
                row_ext_from_stage<N>(reg, launder(t), stg, nyqA, nyqB);
                lds_barrier();
                bool vn = true;
                const int xn = (it + 1 < iters) ? pair_of(it + 1, vn) + half : -1;
                if (xn >= 0) row_dma_issue<N, SLAB>(stg, launder(t), a.M, 0, 1, xn, xn, nyqA, nyqB);
            } else row_load_pair<N, SLAB>(reg, launder(t), a.M, 2, 3, x, x);
            rowfft<N, false>(lds, launder(t), twb, reg);
            float val[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float u = -(reg[e].x * a.scale);                // main.cpp:200-201
                const float v = reg[e].y * a.scale;                   // main.cpp:214
                val[e] = -u * zx[e] - v * zy[e];                      // main.cpp:225-227 ...
            }
            if (a.src) {                                              // ... + vort_src; the loads and their wait stay in this branch
#pragma unroll
                for (int e = 0; e < 16; ++e) val[e] += a.src[(size_t)x * N + t_it + ord_i<RL>(e) * T];
            }
            // both rows' values to both groups: the forward transform packs row x0 (real part) with row x0 + 1 (imaginary part)
            const int tv = launder(t);
            lds_barrier();                                            // (the previous pair's readers are done with the value buffer)
#pragma unroll
            for (int e = 0; e < 16; ++e) valb[(grp * 16 + e) * T + tv] = val[e];
            lds_barrier();
#pragma unroll
            for (int e = 0; e < 16; ++e) reg[e] = cf_make(valb[((grp & ~1) * 16 + e) * T + tv], valb[((grp | 1) * 16 + e) * T + tv]);
            valid = valid && half == 0;                               // the even group stores
        } else if (MODE == ROW_FUSED) {
            float t0[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) t0[e] = 0.f;
#pragma unroll 1
            for (int r = 0; r < 2; ++r) {
                const int x = x0 + r;
                float zx[16], zy[16];
                // ---- phase (r, 0): dvortdx, dvortdy of row x
                if (DMA) {
                    // vmcnt counts stores too, in order: phase (0,0)'s rows were waited for BEFORE the previous
                    // pair's stores went out (below / prologue), so that nobody sits on their acknowledgements
                    if (r == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    lds_barrier();
                    row_ext_from_stage<N>(reg, launder(t), stg, nyqA, nyqB);
                    lds_barrier();
                    row_dma_issue<N, SLAB>(stg, launder(t), a.M, 2, 3, x, x, nyqA, nyqB);
                } else {
                    row_load_pair<N, SLAB>(reg, launder(t), a.M, 0, 1, x, x);
                }
                rowfft<N, false>(lds, launder(t), twb, reg);
#pragma unroll
                for (int e = 0; e < 16; ++e) { zx[e] = reg[e].x * a.scale; zy[e] = reg[e].y * a.scale; }   // main.cpp:154,168
                // ---- phase (r, 1): u, v of row x
                if (DMA) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    lds_barrier();
                    row_ext_from_stage<N>(reg, launder(t), stg, nyqA, nyqB);
                    lds_barrier();
                    bool vn = true; int xn = x1;
                    if (r == 1) xn = (it + 1 < iters) ? pair_of(it + 1, vn) : -1;
                    if (xn >= 0) row_dma_issue<N, SLAB>(stg, launder(t), a.M, 0, 1, xn, xn, nyqA, nyqB);
                } else {
                    row_load_pair<N, SLAB>(reg, launder(t), a.M, 2, 3, x, x);
                }
                rowfft<N, false>(lds, launder(t), twb, reg);
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float u = -(reg[e].x * a.scale);            // main.cpp:200-201
                    const float v = reg[e].y * a.scale;               // main.cpp:214
                    reg[e].y = -u * zx[e] - v * zy[e];                // main.cpp:225-227 ...
                }
                if (a.src) {                                          // ... + vort_src; the loads and their wait stay in this branch
#pragma unroll
                    for (int e = 0; e < 16; ++e) reg[e].y += a.src[(size_t)x * N + t_it + ord_i<RL>(e) * T];
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float val = reg[e].y;
                    reg[e] = cf_make(t0[e], val);      // complete only after r == 1
                    t0[e] = val;
                }
            }
        } else if (MODE == ROW_FWD) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int y = t_it + ord_i<RL>(e) * T;
                reg[e] = cf_make(a.rin[(size_t)x0 * N + y], a.rin[(size_t)x1 * N + y]);
            }
        }

        if (MODE == ROW_FUSED || MODE == ROW_FWD) {
            if constexpr (SHARE) rowfft<N, true>(lds, launder(t), reinterpret_cast<const RowTwSrc<N, true, true> &>(twb), reg);   // main.cpp:237 (y part)
            else rowfft<N, true>(lds, launder(t), twf_own, reg);
            if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the next pair's first rows have landed
            row_store_pair<N, SLAB>(lds, launder(t), reg, valid, a.T, a.t_frozen, x0, x1);
        } else {
            row_load_pair<N, SLAB>(reg, launder(t), a.M, 0, 0, x0, x1);
            rowfft<N, false>(lds, launder(t), twb, reg);
            if (valid) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int y = t_it + ord_i<RL>(e) * T;
                    a.rout[(size_t)x0 * N + y] = reg[e].x * a.scale;
                    a.rout[(size_t)x1 * N + y] = reg[e].y * a.scale;
                }
            }
        }
    }
}

// -------------------------------------------------------------------------------------------
// column sub-passes on wave tiles (16 columns x n rows), 4 waves per workgroup
// -------------------------------------------------------------------------------------------
// Row r of field f of a (possibly destination-blocked) mixed-space array lives at
//   data + (r >> xl_shift)*dstride + f*fstride + (r & xl_mask)*P
// one GPU: xl_shift = 31 (one block), so this is data + f*fstride + r*P.  Slab mode: the four derivative
// fields are laid out [dst rank][field][local row][KS] so that ONE all-to-all moves all of them.
struct RowMap {
    int xl_shift, xl_mask; long dstride;
    int xl;            // rows per destination block when it is not a power of two (3*2^k grids), else 0
    FB_DEV size_t off(int r, int P) const
    {
        if (xl) { const int d = r / xl; return (size_t)d * dstride + (size_t)(r - d * xl) * P; }
        return (size_t)(r >> xl_shift) * dstride + (size_t)(r & xl_mask) * P;
    }
};

struct ColArgs {
    cf *data;          // field f at data + f*fstride (+ row map)
    long fstride;
    RowMap rm;
    int ct0, nct;      // column tiles [ct0, ct0+nct) are processed (frozen high-ky tiles are skipped per stage)
    int pace;          // 1: idle 256 cycles between a wave's consecutive strided accesses (large grids)
    int nfields;
    int P;             // pitch (complex)
    int N1, N2;        // nx = N1*N2
    const cf *tw_n;    // W_n^j, j < n   (n = transform length of this kernel)
    const cf *tw_big;  // W_nx^j, j < nx (block kernels)
};

// strided sub-pass: FFT of length n = N1 over a (rows N2*a + b), in place
template <int n, int DIR>
__global__ void __launch_bounds__(256) k_col_strided(ColArgs a)
{
    using W = WaveTile<n>;
    __shared__ __attribute__((aligned(16))) cf smem[4 * W::LDS_CF];
    const int wv = threadIdx.x >> 6;
    cf *lds = smem + wv * W::LDS_CF;
    const int ntc = a.nct;
    const long ntiles = (long)a.nfields * a.N2 * ntc;
    for (long tile = (long)blockIdx.x * 4 + wv; tile < ntiles; tile += (long)gridDim.x * 4) {
        const int lane = launder((int)(threadIdx.x & 63));
        const int g = lane >> 3, cp = lane & 7, h = lane >> 4, c = lane & 15;
        const int f = (int)(tile / ((long)a.N2 * ntc));
        const int rem = (int)(tile - (long)f * a.N2 * ntc);
        const int b = rem / ntc, ct = a.ct0 + rem - b * ntc;
        cf *base = a.data + (size_t)f * a.fstride + ct * 16;
        float4 in[W::NLA];
#pragma unroll
        for (int m = 0; m < W::NLA; ++m) {
            in[m] = ld4<(FB_NT & 1) != 0 || (FB_NT_FWD && DIR < 0)>(base + a.rm.off((g + 8 * m) * a.N2 + b, a.P) + 2 * cp);
            access_gap(a.pace);
        }
        cf out[W::NLB];
        wave_fft_A2B<n, DIR>(in, out, lds, a.tw_n, lane);
        if (W::lb_active(lane)) {
#pragma unroll
            for (int s = 0; s < W::NP; ++s)
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int k = h + 4 * s + W::R1 * q;
                    st2<(FB_NT & 2) != 0 || (FB_NT_FWD && DIR < 0)>(&base[a.rm.off(k * a.N2 + b, a.P) + c], out[s * 8 + q]);
                    access_gap(a.pace);
                }
        }
    }
}

// block sub-pass: FFT of length n = N2 over the contiguous rows N2*cb + b, with the W_nx^{b cb}
// twiddle (forward: on load; backward: conjugate on store), in place
template <int n, int DIR>
__global__ void __launch_bounds__(256) k_col_block(ColArgs a)
{
    using W = WaveTile<n>;
    __shared__ __attribute__((aligned(16))) cf smem[4 * W::LDS_CF];
    const int wv = threadIdx.x >> 6;
    cf *lds = smem + wv * W::LDS_CF;
    const int ntc = a.nct;
    const long ntiles = (long)a.nfields * a.N1 * ntc;
    for (long tile = (long)blockIdx.x * 4 + wv; tile < ntiles; tile += (long)gridDim.x * 4) {
        const int lane = launder((int)(threadIdx.x & 63));
        const int g = lane >> 3, cp = lane & 7, h = lane >> 4, c = lane & 15;
        const int f = (int)(tile / ((long)a.N1 * ntc));
        const int rem = (int)(tile - (long)f * a.N1 * ntc);
        const int cb = rem / ntc, ct = a.ct0 + rem - cb * ntc;
        cf *base = a.data + (size_t)f * a.fstride + (size_t)cb * n * a.P + ct * 16;
        if (DIR < 0) {
            float4 in[W::NLA];
#pragma unroll
            for (int m = 0; m < W::NLA; ++m) {
                const int b = g + 8 * m;
                float4 v = *reinterpret_cast<const float4 *>(base + (size_t)b * a.P + 2 * cp);
                const cf w = a.tw_big[b * cb];
                cf p0 = cmul(cf_make(v.x, v.y), w), p1 = cmul(cf_make(v.z, v.w), w);
                in[m] = make_float4(p0.x, p0.y, p1.x, p1.y);
            }
            cf out[W::NLB];
            wave_fft_A2B<n, -1>(in, out, lds, a.tw_n, lane);
            if (W::lb_active(lane)) {
#pragma unroll
                for (int s = 0; s < W::NP; ++s)
#pragma unroll
                    for (int q = 0; q < 8; ++q) base[(size_t)(h + 4 * s + W::R1 * q) * a.P + c] = out[s * 8 + q];
            }
        } else {
            cf in[W::NLB];
            if (W::lb_active(lane)) {
#pragma unroll
                for (int s = 0; s < W::NP; ++s)
#pragma unroll
                    for (int q = 0; q < 8; ++q) in[s * 8 + q] = base[(size_t)(h + 4 * s + W::R1 * q) * a.P + c];
            }
            float4 out[W::NLA];
            wave_fft_B2A<n, +1>(in, out, lds, a.tw_n, lane);
#pragma unroll
            for (int m = 0; m < W::NLA; ++m) {
                const int b = g + 8 * m;
                const cf w = a.tw_big[b * cb];
                cf p0 = cmulc(cf_make(out[m].x, out[m].y), w), p1 = cmulc(cf_make(out[m].z, out[m].w), w);
                *reinterpret_cast<float4 *>(base + (size_t)b * a.P + 2 * cp) = make_float4(p0.x, p0.y, p1.x, p1.y);
            }
        }
    }
}

// -------------------------------------------------------------------------------------------
// fused middle kernel: [forward block sub-pass of the tendency] + viscous term + mask + RK4
// update + the four spectral derivatives + [backward block sub-pass of each]
// -------------------------------------------------------------------------------------------
#ifndef FB_MID_REMAKE_ZC   /* 0: stage 1 reads its stage state back (A/B build for the bitwise test) */
#define FB_MID_REMAKE_ZC 1
#endif
struct MidArgs {
    const cf *Tin;       // tendency after the strided forward sub-pass (mixed/partial layout)
    const cf *Zbase;     // vort_c0 of this step                    (private spectral layout)
    cf *Zcur;            // vort_c of this stage; updated in place (stage 0: written only)
    cf *Acc;             // running rk1 + 2 rk2 + 2 rk3
    cf *Zout;            // stage 3: new vort_c is written here (== Zbase's buffer)
    cf *W4;              // four derived fields, field f at W4 + f*fstride (+ row map)
    long fstride;
    RowMap rm;
    int ct0, nct;        // column-tile range
    int P, N1, N2;
    int ky0;             // global ky of local column 0 (slab offset)
    int stage;           // 0..3 RK stage whose tendency arrives; -1 = derive only (prime the pipeline)
    float nu, dt;
    SpecCoef coef;
    const cf *tw_n, *tw_big;
    int split;           // 1: small launch (fewer tiles than the chip has room for): ONE tile per workgroup, every wave repeats the forward
                         // sub-pass and the update, wave 0 stores the state, wave f sends derivative field f through its backward sub-pass --
                         // two dependent transforms per wave instead of five
};

// Memory discipline (gfx9 counts loads AND stores on vmcnt, returned in order): a wave that waits for a
// load issued after a store also waits for that store's acknowledgement.  So no table is read from
// global memory inside the tile: W_n, this tile's W_nx^{b cb}, gradx_coe and its square sit in LDS
// (lgkmcnt), the state arrays are read in half-tile batches (all loads, then the arithmetic, then all
// stores), and the four derivative fields leave with nothing behind them to wait for.
template <int n>
__global__ void __launch_bounds__(256, WaveTile<n>::MID_MIN_WAVES) k_col_mid(MidArgs a)
{
    using W = WaveTile<n>;
    __shared__ __attribute__((aligned(16))) cf smem[4 * W::LDS_CF + n + 4 * n];
    __shared__ float s_gx[4 * n];                    // (double)gradx_coe^2 is formed from it on the fly: exactly the table entry, 4 n doubles of LDS less
    const int wv = threadIdx.x >> 6;
    cf *lds = smem + wv * W::LDS_CF;
    cf *twn = smem + 4 * W::LDS_CF;                  // W_n^k, whole workgroup
    cf *twb = twn + n + wv * n;                      // W_nx^{b cb} of this wave's tile, b = 0..n-1
    float *gxt = s_gx + wv * n;                      // gradx_coe at kx = cb + N1 d, d = 0..n-1
    for (int i = threadIdx.x; i < n; i += 256) twn[i] = a.tw_n[i];
    __syncthreads();
    const int ntc = a.nct, ntc_all = a.P >> 4;
    const long ntiles = (long)a.N1 * ntc;
    const bool split = a.split != 0;                 // (uniform; every wave of a workgroup then runs the same tiles, so the barriers below are met by all)
    for (long wt = split ? (long)blockIdx.x : (long)blockIdx.x * 4 + wv; wt < ntiles; wt += split ? (long)gridDim.x : (long)gridDim.x * 4) {
        // per-tile opaque lane id: keeps address/coefficient arithmetic out of loop-invariant registers
        const int lane = launder((int)(threadIdx.x & 63));
        const int g = lane >> 3, cp = lane & 7, h = lane >> 4, c = lane & 15;
        const bool lb = W::lb_active(lane);
        const int cb = (int)(wt / ntc), ct = a.ct0 + (int)(wt - (long)cb * ntc);
        const long tile = (long)cb * ntc_all + ct;              // index of the tile in the tile-major state arrays
        const size_t tbase = (size_t)cb * n * a.P + ct * 16;
        const int col = ct * 16 + c, ky = a.ky0 + col;
        const float gy = a.coef.gy[ky];
        const double ky2 = a.coef.ky2[ky];
        float4 in[W::NLA];
        if (a.stage >= 0) {
#pragma unroll
            for (int m = 0; m < W::NLA; ++m)
                in[m] = ld4<(FB_NT & 16) != 0>(a.Tin + tbase + (size_t)(g + 8 * m) * a.P + 2 * cp);
        }
        __builtin_amdgcn_wave_barrier();                        // the previous tile's table readers are done
#pragma unroll
        for (int i = 0; i < (n + 63) / 64; ++i) {
            const int d = lane + 64 * i;
            if (n >= 64 || d < n) {
                twb[d] = a.tw_big[d * cb];
                gxt[d] = a.coef.gx[cb + a.N1 * d];
            }
        }
        __builtin_amdgcn_wave_barrier();

        cf zn[W::NLB];                       // state the derivatives are taken of (LB layout)
        if (a.stage >= 0) {
#pragma unroll
            for (int m = 0; m < W::NLA; ++m) {
                const cf wb = twb[g + 8 * m];
                cf p0 = cmul(cf_make(in[m].x, in[m].y), wb), p1 = cmul(cf_make(in[m].z, in[m].w), wb);
                in[m] = make_float4(p0.x, p0.y, p1.x, p1.y);
            }
            cf th[W::NLB];
            wave_fft_A2B<n, -1>(in, th, lds, twn, lane);               // main.cpp:237 (x part)
            if (lb) {
                // one RK update of element e; dvortdt_c += lvort_c * NU ; rk = dealiase(dvortdt_c)   main.cpp:148,240-243,296
                auto update = [&](int e, cf z0, cf zcur, cf ac, cf &acc, cf &znew) {
                    const int d = h + 4 * (e >> 3) + W::R1 * (e & 7), ikx = cb + a.N1 * d;
                    const float lap = (float)(-((double)gxt[d] * (double)gxt[d] + ky2));     // fftwfop.cpp:42,45
                    const float msk = coef_mask(a.coef, ikx, ky);
                    const cf zc = a.stage == 0 ? z0 : zcur;
                    cf k = cf_make((th[e].x + (zc.x * lap) * a.nu) * msk, (th[e].y + (zc.y * lap) * a.nu) * msk);
                    if (a.stage == 0) {            // main.cpp:296   (explicit fma: stage 1 forms this value again from the stored accumulator)
                        acc = k; znew = cf_make(__builtin_fmaf(k.x, a.dt / 2.0f, z0.x), __builtin_fmaf(k.y, a.dt / 2.0f, z0.y));
                    } else if (a.stage == 1) {     // main.cpp:299
                        acc = cf_make(ac.x + 2.0f * k.x, ac.y + 2.0f * k.y);
                        znew = cf_make(z0.x + k.x * (a.dt / 2.0f), z0.y + k.y * (a.dt / 2.0f));
                    } else if (a.stage == 2) {     // main.cpp:302
                        acc = cf_make(ac.x + 2.0f * k.x, ac.y + 2.0f * k.y);
                        znew = cf_make(z0.x + k.x * a.dt, z0.y + k.y * a.dt);
                    } else {                       // main.cpp:309-312
                        acc = ac;
                        znew = cf_make(z0.x + (ac.x + k.x) * a.dt / 6.0f, z0.y + (ac.y + k.y) * a.dt / 6.0f);
                    }
                };
                if (W::TM) {
                    // state arrays in the tile-major layout (see k_state_relayout): one float4 = elements e, e+1
                    constexpr int JH = W::NLB / 4;              // float4 per lane and array in a half-tile batch
                    const size_t sb = ((size_t)tile * (W::NLB / 2)) * 64 + lane;
#pragma unroll
                    for (int hb = 0; hb < 2; ++hb) {
                        float4 q0[JH], q1[JH], q2[JH];
                        bool frozen[JH];
                        // modes outside the dealiasing circle never change (SURVEY note N1): when all 128 elements
                        // of a wave instruction are masked, the stage state and the accumulator are not touched
#pragma unroll
                        for (int j = 0; j < JH; ++j) {
                            const int jp = hb * JH + j;
                            const int d0 = h + 4 * ((2 * jp) >> 3) + W::R1 * ((2 * jp) & 7), d1 = h + 4 * ((2 * jp + 1) >> 3) + W::R1 * ((2 * jp + 1) & 7);
                            const bool mine = coef_mask(a.coef, cb + a.N1 * d0, ky) == 0.0f && coef_mask(a.coef, cb + a.N1 * d1, ky) == 0.0f;
                            frozen[j] = __all(mine);
                        }
#pragma unroll
                        for (int j = 0; j < JH; ++j) {
                            const int jp = hb * JH + j;
                            q0[j] = ld4<(FB_NT & 64) != 0>(reinterpret_cast<const float4 *>(a.Zbase) + sb + jp * 64);
                            q1[j] = q2[j] = make_float4(0.f, 0.f, 0.f, 0.f);            // never looked at when masked, or at stage 0
                            if (!frozen[j] && a.stage != 0) {
                                q2[j] = ld4<(FB_NT & 64) != 0>(reinterpret_cast<const float4 *>(a.Acc) + sb + jp * 64);
                                // stage 1 does not read its stage state back: stage 0 stored fma(rk1, dt/2, vort_c0) and, next to it, rk1 as
                                // the accumulator -- the same instruction on the same bits gives it again (as in k_col_full)
                                if (a.stage != 1 || !FB_MID_REMAKE_ZC) q1[j] = ld4<(FB_NT & 64) != 0>(reinterpret_cast<const float4 *>(a.Zcur) + sb + jp * 64);
                            }
                        }
#pragma unroll
                        for (int j = 0; j < JH; ++j) {
                            const int e = 2 * (hb * JH + j);
                            cf acc0, acc1;
                            float4 zq = frozen[j] ? q0[j] : q1[j];
                            if (FB_MID_REMAKE_ZC && !frozen[j] && a.stage == 1) {
                                const float hd = a.dt / 2.0f;
                                zq = make_float4(__builtin_fmaf(q2[j].x, hd, q0[j].x), __builtin_fmaf(q2[j].y, hd, q0[j].y),
                                                 __builtin_fmaf(q2[j].z, hd, q0[j].z), __builtin_fmaf(q2[j].w, hd, q0[j].w));
                            }
                            update(e, cf_make(q0[j].x, q0[j].y), cf_make(zq.x, zq.y), cf_make(q2[j].x, q2[j].y), acc0, zn[e]);
                            update(e + 1, cf_make(q0[j].z, q0[j].w), cf_make(zq.z, zq.w), cf_make(q2[j].z, q2[j].w), acc1, zn[e + 1]);
                            q2[j] = make_float4(acc0.x, acc0.y, acc1.x, acc1.y);
                        }
                        if (split) __syncthreads();          // every wave has read the old state before wave 0 overwrites it
#pragma unroll
                        for (int j = 0; j < JH; ++j) {
                            const int jp = hb * JH + j, e = 2 * jp;
                            if (frozen[j] || (split && wv != 0)) continue;
                            const float4 zo = make_float4(zn[e].x, zn[e].y, zn[e + 1].x, zn[e + 1].y);
                            if (a.stage < 3) {
                                st4<(FB_NT & 64) != 0>(reinterpret_cast<float4 *>(a.Acc) + sb + jp * 64, q2[j]);
                                st4<(FB_NT & 64) != 0>(reinterpret_cast<float4 *>(a.Zcur) + sb + jp * 64, zo);
                            } else st4<(FB_NT & 64) != 0>(reinterpret_cast<float4 *>(a.Zout) + sb + jp * 64, zo);
                        }
                    }
                } else {
                    cf z0[W::NLB], zc[W::NLB], ac[W::NLB];
#pragma unroll
                    for (int e = 0; e < W::NLB; ++e) {
                        const size_t off = tbase + (size_t)(h + 4 * (e >> 3) + W::R1 * (e & 7)) * a.P + c;
                        z0[e] = a.Zbase[off];
                        if (a.stage != 0) { zc[e] = a.Zcur[off]; ac[e] = a.Acc[off]; } else { zc[e] = z0[e]; ac[e] = cf_make(0.f, 0.f); }
                    }
#pragma unroll
                    for (int e = 0; e < W::NLB; ++e) update(e, z0[e], zc[e], ac[e], ac[e], zn[e]);
                    if (split) __syncthreads();              // (every wave has lanes in here) the old state has been read by all before wave 0 overwrites it
#pragma unroll
                    for (int e = 0; e < W::NLB; ++e) {
                        const size_t off = tbase + (size_t)(h + 4 * (e >> 3) + W::R1 * (e & 7)) * a.P + c;
                        if (split && wv != 0) continue;
                        if (a.stage < 3) { a.Acc[off] = ac[e]; a.Zcur[off] = zn[e]; }
                        else a.Zout[off] = zn[e];
                    }
                }
            }
        } else if (lb) {
            if (W::TM) {
                const size_t sb = ((size_t)tile * (W::NLB / 2)) * 64 + lane;
#pragma unroll
                for (int jp = 0; jp < W::NLB / 2; ++jp) {
                    const float4 t0 = reinterpret_cast<const float4 *>(a.Zbase)[sb + jp * 64];
                    zn[2 * jp] = cf_make(t0.x, t0.y); zn[2 * jp + 1] = cf_make(t0.z, t0.w);
                }
            } else {
#pragma unroll
                for (int s = 0; s < W::NP; ++s)
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        zn[s * 8 + q] = a.Zbase[tbase + (size_t)(h + 4 * s + W::R1 * q) * a.P + c];
            }
            // land the loads here: otherwise the wait sits in front of the derivative loop for every
            // path, and the stage path would wait for its state stores there
#pragma unroll
            for (int e = 0; e < W::NLB; ++e) asm volatile("" :: "v"(zn[e].x), "v"(zn[e].y));
        }

        // derivatives of zn through the backward block sub-pass: f0 gradx(vort), f1 grady(vort),
        // then zn <- psi = invertLaplacian(vort) in place, f2 grady(psi), f3 gradx(psi)
#pragma unroll 1
        for (int f = split ? wv : 0; f < (split ? wv + 1 : 4); ++f) {
            const int lf = launder(lane);
            const int gf = lf >> 3, cpf = lf & 7, hf = lf >> 4;
            cf fld[W::NLB];
            if (lb) {
                if (f == 2 || (split && f == 3)) {
#pragma unroll
                    for (int s = 0; s < W::NP; ++s)
#pragma unroll
                        for (int q = 0; q < 8; ++q) {                 // main.cpp:179, fftwfop.cpp:112-117
                            const int e = s * 8 + q, d = hf + 4 * s + W::R1 * q, ikx = cb + a.N1 * d;
                            const float li = (ikx == 0 && ky == 0) ? 1.0f : (float)(-((double)gxt[d] * (double)gxt[d] + ky2));
                            zn[e] = (ky < a.coef.hy) ? cf_make(zn[e].x / li, zn[e].y / li) : cf_make(0.f, 0.f);
                        }
                }
                const bool use_gx = (f == 0 || f == 3);
#pragma unroll
                for (int s = 0; s < W::NP; ++s)
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int e = s * 8 + q, d = hf + 4 * s + W::R1 * q;
                        const float kk = use_gx ? gxt[d] : gy;
                        fld[e] = cf_make(-zn[e].y * kk, zn[e].x * kk);                // fftwfop.cpp:87-103
                    }
            }
            float4 out[W::NLA];
            wave_fft_B2A<n, +1>(fld, out, lds, twn, lf);
            cf *dst = a.W4 + (size_t)f * a.fstride + a.rm.off(cb * n, a.P) + ct * 16;      // a block of n rows never straddles a rank
#pragma unroll
            for (int m = 0; m < W::NLA; ++m) {
                const cf wb = twb[gf + 8 * m];
                cf p0 = cmulc(cf_make(out[m].x, out[m].y), wb), p1 = cmulc(cf_make(out[m].z, out[m].w), wb);
                st4<(FB_NT & 32) != 0>(dst + (size_t)(gf + 8 * m) * a.P + 2 * cpf, make_float4(p0.x, p0.y, p1.x, p1.y));
            }
        }
    }
}
