// fb_row3.h -- row pass for ny = 3*M (M a power of two): the reference's shipped default grid is
// NPTS = 768 = 3*2^8 (configuration.hpp:18).
//
// A length-3M complex transform is one radix-3 step plus three length-M transforms, which are the
// power-of-two workgroup FFTs of fb_fft_core.h run by three thread groups side by side:
//   backward (c2r): G_r[k0] = W_N^{-r k0} * sum_j W_3^{-r j} Z[k0 + M j]     (radix-3 on the input thirds)
//                   z[3m + r] = IFFT_M(G_r)[m]                               (group r, r = 0,1,2)
//   forward  (r2c): F_r = FFT_M(z[3m + r]);  Z[k0 + M j] = sum_r W_3^{r j} W_N^{r k0} F_r[k0]
// Same packing as k_row: two real-output transforms per complex backward FFT (Hermitian extension
// with the reference's c2r semantics, SURVEY note N2), two x rows per forward FFT, untangle on store.
// Correctness-first (plain global loads, no LDS-DMA): these grids are not benchmark configurations.
#pragma once
#include "fb_kernels.h"

template <int M> struct Row3Cfg {
    static constexpr int N = 3 * M, T = M / 16, PT = 3 * T;
    // M = 256 (768^2, the reference's default): TWO groups of PT = 48 working threads per workgroup, each on a wave of its own.  In the
    // fused pass they take the two x rows of ONE row pair side by side (the pair's backward transforms used to run one row after the
    // other in one group: five dependent transforms per workgroup on a grid with 384 row pairs for 256 CUs); elsewhere they are two
    // independent pairs.  Larger M has at least as many row pairs as the chip has room for (measured at 1536^2: 0.069 ms with one
    // group per workgroup, 0.101 ms with two), smaller M packs as many pairs as fit 64 threads.
    static constexpr bool TWO = M == 256;
    static constexpr int HS = TWO ? 64 : PT;                    // threads per group
    static constexpr int GP = TWO ? 2 : (PT >= 48 ? 1 : 64 / PT);   // groups per workgroup
    static constexpr int THREADS = GP * HS;
    static constexpr int VALB = TWO ? 16 * PT : 0;              // complex: the two rows' tendency values handed between the groups (2 x 16 x PT floats)
    static constexpr int LSTR = M + M / 16;                 // padded complex per sub-transform buffer
    static constexpr bool SHARE = RowPlanSymmetric<M>::value;
    static constexpr int TWL_B = RowTwSrc<M, false, true>::LDS_CF, TWL_F = SHARE ? 0 : RowTwSrc<M, true, true>::LDS_CF;
    // fused mode: the four half-spectrum rows of an x row staged in LDS (coalesced 16-byte loads, once) instead of every sub-transform
    // group gathering all three thirds of both rows from global memory, 8 bytes at a time
    static constexpr int HP = N / 2 + 2;                    // staged row: X[0 .. N/2] + one pad element (even length)
    static constexpr int STG = 4 * HP;
    static constexpr size_t LDS_BYTES = ((size_t)GP * 3 * LSTR + TWL_B + TWL_F + (size_t)GP * STG + VALB) * sizeof(cf);
};
#ifndef FB_ROW3_STAGE
#define FB_ROW3_STAGE 1
#endif

// the four fields' half-spectrum rows of x row `row` -> stg[f * HP + k]; q = thread within the pair's PT threads
template <int M, bool SLAB>
FB_DEV void row3_stage4(cf *stg, int q, const RowView &v, int row)
{
    using C = Row3Cfg<M>;
#pragma unroll
    for (int f = 0; f < 4; ++f)
        for (int i = q; i < C::HP / 2; i += C::PT) {
            const float4 x = *reinterpret_cast<const float4 *>(row_ptr<SLAB>(v, f, row, 2 * i));     // columns 2i, 2i+1 (the last pair: N/2 and a zero pad column)
            *reinterpret_cast<float4 *>(&stg[f * C::HP + 2 * i]) = x;
        }
}
template <int N> FB_DEV cf row3_z_lds(const cf *sa, const cf *sb, int k)
{
    const bool mirror = 2 * k > N;
    const int kk = mirror ? N - k : k;
    const cf a = sa[kk], b = sb[kk];
    if (kk == 0 || 2 * kk == N) return cf_make(a.x, b.x);                     // Im ignored at k = 0 and k = N/2
    return mirror ? cf_make(a.x + b.y, b.x - a.y) : cf_make(a.x - b.y, a.y + b.x);
}
template <int M>
FB_DEV void row3_load_lds(cf *reg, int r, int t, const cf *sa, const cf *sb, const cf *__restrict__ twN)
{
    constexpr int N = 3 * M, T = M / 16, R0 = RowTw<M, false>::radix(0);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int p0 = t + ord_i<R0>(e) * T;
        cf z0 = row3_z_lds<N>(sa, sb, p0), z1 = row3_z_lds<N>(sa, sb, p0 + M), z2 = row3_z_lds<N>(sa, sb, p0 + 2 * M);
        fft3<+1>(z0, z1, z2);
        const cf g = r == 0 ? z0 : (r == 1 ? z1 : z2);
        reg[e] = r == 0 ? g : cmulc(g, twN[r * p0]);
    }
}

// Hermitian-extended packed spectrum value Z[k], 0 <= k < N, of the two half-spectrum rows A, B
template <int N, bool SLAB>
FB_DEV cf row3_z(const RowView &v, int fA, int fB, int rowA, int rowB, int k)
{
    const bool mirror = 2 * k > N;
    const int kk = mirror ? N - k : k;
    const cf a = *row_ptr<SLAB>(v, fA, rowA, kk), b = *row_ptr<SLAB>(v, fB, rowB, kk);
    if (kk == 0 || 2 * kk == N) return cf_make(a.x, b.x);                     // Im ignored at k = 0 and k = N/2
    return mirror ? cf_make(a.x + b.y, b.x - a.y) : cf_make(a.x - b.y, a.y + b.x);
}

// backward input of sub-transform r: radix-3 over the spectrum thirds + twiddle, into the first stage's registers
template <int M, bool SLAB>
FB_DEV void row3_load(cf *reg, int r, int t, const RowView &v, int fA, int fB, int rowA, int rowB, const cf *__restrict__ twN)
{
    constexpr int N = 3 * M, T = M / 16, R0 = RowTw<M, false>::radix(0);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int p0 = t + ord_i<R0>(e) * T;
        cf z0 = row3_z<N, SLAB>(v, fA, fB, rowA, rowB, p0);
        cf z1 = row3_z<N, SLAB>(v, fA, fB, rowA, rowB, p0 + M);
        cf z2 = row3_z<N, SLAB>(v, fA, fB, rowA, rowB, p0 + 2 * M);
        fft3<+1>(z0, z1, z2);                                   // z_r = sum_j Z[p0 + M j] exp(+2 pi i r j / 3)
        const cf g = r == 0 ? z0 : (r == 1 ? z1 : z2);
        reg[e] = r == 0 ? g : cmulc(g, twN[r * p0]);            // * W_N^{-r p0}
    }
}

// forward output: twiddle, radix-3 combine across the three groups through LDS, untangle, store rows A, B of T
template <int M, bool SLAB>
FB_DEV void row3_store(cf *lds_pair, int r, int t, const cf *reg, bool valid, const RowView &v, int t_frozen, int rowA, int rowB,
                       const cf *__restrict__ twN)
{
    constexpr int N = 3 * M, T = M / 16, LSTR = Row3Cfg<M>::LSTR, RL = RowTw<M, true>::radix(RowPlan<M>::S - 1);
    lds_barrier();                                              // the last stage's readers are done
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int k0 = t + ord_i<RL>(e) * T;
        lds_pair[r * LSTR + lds_pad(k0)] = r == 0 ? reg[e] : cmul(reg[e], twN[r * k0]);     // W_N^{r k0} F_r[k0]
    }
    lds_barrier();
    auto zval = [&](int k) {                                    // Z[k] = sum_r W_3^{r j} CB[r][k0], k = k0 + M j
        const int j = k / M, k0 = k - j * M;
        cf c0 = lds_pair[lds_pad(k0)], c1 = lds_pair[LSTR + lds_pad(k0)], c2 = lds_pair[2 * LSTR + lds_pad(k0)];
        fft3<-1>(c0, c1, c2);
        return j == 0 ? c0 : (j == 1 ? c1 : c2);
    };
    if (valid) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = (r * 8 + i) * T + t;                  // all k in [0, N/2)
            const cf zk = zval(k), zn = k == 0 ? zk : zval(N - k);
            if (!row_keep<SLAB>(v, t_frozen, k)) continue;
            *const_cast<cf *>(row_ptr<SLAB>(v, 0, rowA, k)) = cf_make(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y));
            *const_cast<cf *>(row_ptr<SLAB>(v, 0, rowB, k)) = cf_make(0.5f * (zk.y + zn.y), 0.5f * (zn.x - zk.x));
        }
        if (r == 0 && t == 0 && row_keep<SLAB>(v, t_frozen, N / 2)) {      // Nyquist: its own mirror
            const cf z = zval(N / 2);
            *const_cast<cf *>(row_ptr<SLAB>(v, 0, rowA, N / 2)) = cf_make(z.x, 0.f);
            *const_cast<cf *>(row_ptr<SLAB>(v, 0, rowB, N / 2)) = cf_make(z.y, 0.f);
        }
    }
}

template <int M, int MODE, bool SLAB>
__global__ void __launch_bounds__(Row3Cfg<M>::THREADS) k_row3(RowArgs a, const cf *__restrict__ twN)
{
    using C = Row3Cfg<M>;
    constexpr int N = C::N, T = C::T, PT = C::PT, GP = C::GP;
    constexpr int RL = RowTw<M, false>::radix(RowPlan<M>::S - 1);      // physical-space register order
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf *smem = reinterpret_cast<cf *>(smem_raw);
    constexpr bool PAIR2 = C::TWO && MODE == ROW_FUSED;        // the two groups work on the two rows of one pair
    constexpr int PPW = PAIR2 ? 1 : GP;                         // row pairs per workgroup
    const int tid = threadIdx.x, gp = tid / C::HS, q0 = tid - gp * C::HS;
    const bool active = q0 < PT;                                // (groups sit on whole waves: the lanes beyond PT repeat thread 0's work and store nothing)
    const int q = active ? q0 : 0, r = q / T, t = q - r * T;
    cf *lds_pair = smem + (size_t)gp * 3 * C::LSTR;
    cf *lds = lds_pair + r * C::LSTR;
    cf *twl = smem + (size_t)GP * 3 * C::LSTR;
    cf *stg = twl + C::TWL_B + C::TWL_F + (size_t)gp * C::STG;
    float *valb = reinterpret_cast<float *>(twl + C::TWL_B + C::TWL_F + (size_t)GP * C::STG);
    RowTwSrc<M, false, true> twb;
    twb.init(a.tw_bwd, twl, t, tid, C::THREADS);
    RowTwSrc<M, true, true> twf_own;
    if (!C::SHARE) twf_own.init(a.tw_fwd, twl + C::TWL_B, t, tid, C::THREADS);
    __syncthreads();

    const int npairs = a.nx >> 1;
    const int iters = (npairs + gridDim.x * PPW - 1) / (gridDim.x * PPW);
    for (int it = 0; it < iters; ++it) {
        const int pr = (it * gridDim.x + blockIdx.x) * PPW + (PAIR2 ? 0 : gp);
        const bool valid = pr < npairs && active && (!PAIR2 || gp == 0);     // who stores
        const int x0 = a.x0 + (pr < npairs ? 2 * pr : 0), x1 = x0 + 1;
        const int tl = launder(t);
        cf reg[16];
        if (PAIR2) {
            // this group's row of the pair: four backward transforms' worth of work split over the two groups
            const int x = x0 + gp;
            float zx[16], zy[16], val[16];
            __syncthreads();                                      // the previous pair's readers are done with the staged rows and the value buffer
            row3_stage4<M, SLAB>(stg, launder(q), a.M, x);
            __syncthreads();
            row3_load_lds<M>(reg, r, launder(t), stg, stg + C::HP, twN);
            row_fft<M, false>(lds, launder(t), twb, reg);
#pragma unroll
            for (int e = 0; e < 16; ++e) { zx[e] = reg[e].x * a.scale; zy[e] = reg[e].y * a.scale; }   // main.cpp:154,168
            row3_load_lds<M>(reg, r, launder(t), stg + 2 * C::HP, stg + 3 * C::HP, twN);
            row_fft<M, false>(lds, launder(t), twb, reg);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float u = -(reg[e].x * a.scale);            // main.cpp:200-201
                const float v = reg[e].y * a.scale;               // main.cpp:214
                const int y = 3 * (tl + ord_i<RL>(e) * T) + r;
                const float s = a.src ? a.src[(size_t)x * N + y] : 0.0f;
                val[e] = -u * zx[e] - v * zy[e] + s;              // main.cpp:225-227
            }
            // both rows' values to both groups: the forward transform packs row x0 (real part) and row x0 + 1 (imaginary part); both
            // groups run it (a barrier inside it must be met by every wave), group 0 stores
            const int ql = launder(q);
#pragma unroll
            for (int e = 0; e < 16; ++e) valb[(gp * 16 + e) * PT + ql] = val[e];
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 16; ++e) reg[e] = cf_make(valb[e * PT + ql], valb[(16 + e) * PT + ql]);
        } else if (MODE == ROW_FUSED) {
            float t0[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) t0[e] = 0.f;
#pragma unroll 1
            for (int rr = 0; rr < 2; ++rr) {
                const int x = x0 + rr;
                float zx[16], zy[16];
                if (FB_ROW3_STAGE) {
                    __syncthreads();                                  // the previous row's readers are done with the staged rows
                    row3_stage4<M, SLAB>(stg, launder(q), a.M, x);
                    __syncthreads();
                    row3_load_lds<M>(reg, r, launder(t), stg, stg + C::HP, twN);
                } else row3_load<M, SLAB>(reg, r, launder(t), a.M, 0, 1, x, x, twN);
                row_fft<M, false>(lds, launder(t), twb, reg);
#pragma unroll
                for (int e = 0; e < 16; ++e) { zx[e] = reg[e].x * a.scale; zy[e] = reg[e].y * a.scale; }   // main.cpp:154,168
                if (FB_ROW3_STAGE) row3_load_lds<M>(reg, r, launder(t), stg + 2 * C::HP, stg + 3 * C::HP, twN);
                else row3_load<M, SLAB>(reg, r, launder(t), a.M, 2, 3, x, x, twN);
                row_fft<M, false>(lds, launder(t), twb, reg);
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float u = -(reg[e].x * a.scale);            // main.cpp:200-201
                    const float v = reg[e].y * a.scale;               // main.cpp:214
                    const int y = 3 * (tl + ord_i<RL>(e) * T) + r;
                    const float s = a.src ? a.src[(size_t)x * N + y] : 0.0f;
                    const float val = -u * zx[e] - v * zy[e] + s;     // main.cpp:225-227
                    reg[e] = cf_make(t0[e], val);
                    t0[e] = val;
                }
            }
        } else if (MODE == ROW_FWD) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int y = 3 * (tl + ord_i<RL>(e) * T) + r;
                reg[e] = cf_make(a.rin[(size_t)x0 * N + y], a.rin[(size_t)x1 * N + y]);
            }
        }
        if (MODE == ROW_FUSED || MODE == ROW_FWD) {
            if constexpr (C::SHARE) row_fft<M, true>(lds, launder(t), reinterpret_cast<const RowTwSrc<M, true, true> &>(twb), reg);
            else row_fft<M, true>(lds, launder(t), twf_own, reg);
            row3_store<M, SLAB>(lds_pair, r, launder(t), reg, valid, a.T, a.t_frozen, x0, x1, twN);
        } else {
            row3_load<M, SLAB>(reg, r, launder(t), a.M, 0, 0, x0, x1, twN);
            row_fft<M, false>(lds, launder(t), twb, reg);
            if (valid) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int y = 3 * (tl + ord_i<RL>(e) * T) + r;
                    a.rout[(size_t)x0 * N + y] = reg[e].x * a.scale;
                    a.rout[(size_t)x1 * N + y] = reg[e].y * a.scale;
                }
            }
        }
    }
}
