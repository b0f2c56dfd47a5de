// fb_rowh.h -- fused row pass for long rows: ny = 8192 (V = 1) and ny = 16384 (V = 2), 512 threads per x row.
//
// Why a third row kernel: the Stockham kernel (k_row) packs two real rows into one complex transform of ny points, which at
// ny = 8192 / 16384 needs 64 / 128 KB of exchange LDS plus as much again for prefetching -- one workgroup per CU, 2 waves per
// SIMD, and the HBM, LDS and ALU phases of that one workgroup do not overlap (measured: 3.0 and 2.3 TB/s of its 5 C of traffic).
// Here ONE real row of N = ny points is one complex transform of M = N/2 points (even/odd packing):
//     c2r:  Z[k] = (X[k] + conj X[M-k]) + i e^{+2 pi i k/N} (X[k] - conj X[M-k]),  z = IDFT_M(Z),  x[2j] + i x[2j+1] = z[j]
//     r2c:  w[j] = t[2j] + i t[2j+1],  W = DFT_M(w),  E = (W[k] + conj W[M-k])/2,  O = (W[k] - conj W[M-k])/(2i),
//           T[k] = E + e^{-2 pi i k/N} O,  T[M-k] = conj(E - e^{-2 pi i k/N} O),  T[0] = Re W0 + Im W0,  T[M] = Re W0 - Im W0
// (the imaginary parts of X[0] and X[M] are ignored, as FFTW's c2r does: SURVEY.md note N2).  M = 4096 V: V interleaved
// sub-sequences of 4096 = 8^4 points, each transformed with fb_row8.h's digit scheme (registers / wave / l_hi / l_lo), and for
// V = 2 one radix-2 step in registers (decimation in time going backward, in frequency going forward).  So the working set
// per workgroup is 37 V KB of exchange LDS + one staged half-spectrum row (32 V KB): two workgroups per CU at ny = 8192, one
// (with 16 registers' worth of transform per thread) at ny = 16384.
//
// Per x row: four c2r transforms (d vort/dx, u, d vort/dy, v -- in that order, so that only two physical-space arrays are
// ever live: -u * dvortdx is formed as soon as u arrives), the Jacobian (main.cpp:225-227), one r2c transform.  The next
// transform's row travels by LDS-DMA while the current one runs.  The physical-space order is digit-reversed (it never
// matters: the Jacobian is pointwise; vort_src is permuted once into that order by k_rowh_permute_src).
// Same arithmetic as k_row<N, ROW_FUSED> up to rounding order (main.cpp:154-237, y part).
#pragma once
#include "fb_row8.h"

// Opaque thread id per phase (launder, see fb_rowq.h): needed where the registers are short (V = 1: 122 of 128); at V = 2 the kernel runs two
// waves per SIMD with up to 256 registers and the one-GPU instance can keep its LDS addresses (246, no scratch; the slab-blocked one would spill
// 600 B).  RH_LAUNDER_ALL=1: always.
#ifndef RH_LAUNDER_ALL
#define RH_LAUNDER_ALL 0
#endif
#define RH_LAUNDER(t) ((V == 1 || SLAB || RH_LAUNDER_ALL) ? launder(t) : (t))
#ifndef RH_NT      /* nontemporal hint on the LDS-DMA loads of the four fields (as RQ_NT in fb_rowq.h) */
#define RH_NT 0
#endif
template <int V> struct RowH {
    static constexpr int M = 4096 * V, N = 2 * M, T = 512;
    static constexpr int SLICE = Row8::SLICE;                  // complex per (wave, sub-sequence) slice of the exchange buffer
    static constexpr int XSUB = 8 * SLICE;                      // one sub-sequence's exchange buffer (8 waves)
    static constexpr int XBUF = V * XSUB;
    static constexpr int STG = M + 2;                           // staged row: X[0..M] (+1 pad)
    static constexpr int TW2 = 64;                              // W_64^{p l_lo} at [p][l_lo]
    static constexpr size_t LDS_BYTES = (size_t)(XBUF + STG + TW2) * sizeof(cf);
    static constexpr int MIN_WAVES = V == 1 ? 4 : 2;            // waves per SIMD the register allocation must allow
};

// ---- the M-point transforms on v[V][8]: sub-sequence s holds Z[V q + s], q = t + 512 e (natural order) ----------------
struct RhNothing { FB_DEV void operator()() const {} };
template <int V, bool LEAD = true, class F = RhNothing> FB_DEV void rh_xch_group(cf (*v)[8], cf *xbuf, int w, int l, F &&behind_barrier = F())
{
#ifdef FB_R8_NOXG   /* timing experiment only (results are wrong) */
    behind_barrier();
    return;
#endif
    if (LEAD) lds_barrier();                          // every wave is done with its slices (not needed by the backward transforms: r8_xch_group)
#pragma unroll
    for (int s = 0; s < V; ++s)
#pragma unroll
        for (int p = 0; p < 8; ++p) lds_wr(&xbuf[s * RowH<V>::XSUB + p * Row8::SLICE + w * 64 + l], v[s][p]);
    lds_barrier();
    behind_barrier();                                 // (backward: every wave has also left the staged row behind -- send for the next one)
#pragma unroll
    for (int s = 0; s < V; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) v[s][e] = lds_rd(&xbuf[s * RowH<V>::XSUB + w * Row8::SLICE + e * 64 + l]);
}

// The wave-private exchanges of the V sub-sequences, issued TOGETHER: all writes, one scheduling fence, all reads.  The sub-sequences
// are independent (slices of their own), so the LDS round trip of one hides behind the other's instead of being paid V times, and
// the butterflies that follow have two independent chains to schedule (a wave has one partner on its SIMD at ny = 16384).
#ifndef RH_INTERLEAVE
#define RH_INTERLEAVE 2
#endif
// one sub-sequence's wave-private exchange in two halves (RH_INTERLEAVE == 2: the other sub-sequence's butterfly is issued between
// them, so that the LDS round trip of one hides behind the arithmetic of the other inside the same wave)
template <int V, bool HI> FB_DEV void rh_xw_write(const cf *v, cf *xbuf, int s, int w, int l_hi, int l_lo)
{
    constexpr int PITCH = HI ? Row8::PITCH_HI : Row8::PITCH_LO;
    cf *wr = xbuf + s * RowH<V>::XSUB + w * Row8::SLICE + l_hi * 8 + l_lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) lds_wr(&wr[e * PITCH], v[e]);
}
template <int V, bool HI> FB_DEV void rh_xw_read(cf *v, const cf *xbuf, int s, int w, int l_hi, int l_lo)
{
    constexpr int PITCH = HI ? Row8::PITCH_HI : Row8::PITCH_LO;
    const cf *slice = xbuf + s * RowH<V>::XSUB + w * Row8::SLICE;
    const cf *rd = HI ? slice + l_hi * PITCH + l_lo : slice + l_lo * PITCH + l_hi * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = lds_rd(&rd[e * (HI ? 8 : 1)]);
}
template <int V, bool HI> FB_DEV void rh_xch_wave_all(cf (*v)[8], cf *xbuf, int w, int l_hi, int l_lo)
{
#ifdef FB_R8_NOXW   /* timing experiment only (results are wrong) */
    return;
#endif
    constexpr int PITCH = HI ? Row8::PITCH_HI : Row8::PITCH_LO;
#pragma unroll
    for (int s = 0; s < V; ++s) {
        cf *wr = xbuf + s * RowH<V>::XSUB + w * Row8::SLICE + l_hi * 8 + l_lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) lds_wr(&wr[e * PITCH], v[s][e]);
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int s = 0; s < V; ++s) {
        const cf *slice = xbuf + s * RowH<V>::XSUB + w * Row8::SLICE;
        const cf *rd = HI ? slice + l_hi * PITCH + l_lo : slice + l_lo * PITCH + l_hi * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[s][e] = lds_rd(&rd[e * (HI ? 8 : 1)]);
    }
    __builtin_amdgcn_wave_barrier();
}

// backward: natural order in, digit-reversed out: sub-transform output F_s[j'] with j' = w + 8 l_hi + 64 l_lo + 512 e; then
// (V = 2) z[j'] = F_0 + W^{-j'} F_1 and z[j' + 4096] = F_0 - W^{-j'} F_1 with W = exp(-2 pi i/M), left in v[0][e], v[1][e]
template <int V, class F> FB_DEV void rh_bwd(cf (*v)[8], cf *xbuf, const Row8Tw &tw, cf wq, int w, int l, F &&behind_barrier)
{
    const int l_hi = l >> 3, l_lo = l & 7;
#if !defined(FB_R8_NOXG)
    if (V > 1 && RH_INTERLEAVE == 2) {                    // the group exchange with sub-sequence s's values leaving while s + 1 is transformed
#pragma unroll
        for (int s = 0; s < V; ++s) {
            Bfly<8, +1>::run(v[s]);
#pragma unroll
            for (int p = 1; p < 8; ++p) v[s][p] = cmulc(v[s][p], tw.w0[p - 1]);
#pragma unroll
            for (int p = 0; p < 8; ++p) lds_wr(&xbuf[s * RowH<V>::XSUB + p * Row8::SLICE + w * 64 + l], v[s][p]);
        }
        lds_barrier();
        behind_barrier();
#pragma unroll
        for (int s = 0; s < V; ++s)
#pragma unroll
            for (int e = 0; e < 8; ++e) v[s][e] = lds_rd(&xbuf[s * RowH<V>::XSUB + w * Row8::SLICE + e * 64 + l]);
    } else
#endif
    {
#pragma unroll
    for (int s = 0; s < V; ++s) {
        Bfly<8, +1>::run(v[s]);
#pragma unroll
        for (int p = 1; p < 8; ++p) v[s][p] = cmulc(v[s][p], tw.w0[p - 1]);
    }
    rh_xch_group<V, false>(v, xbuf, w, l, behind_barrier);
    }
    if (V > 1 && RH_INTERLEAVE == 2) {
        cf w2[7];
#pragma unroll
        for (int p = 1; p < 8; ++p) w2[p - 1] = lds_rd(&tw.w2[p * 8 + l_lo]);
#pragma unroll
        for (int s = 0; s < V; ++s) {                  // write and send for s while s + 1 is still being transformed
            Bfly<8, +1>::run(v[s]);
#pragma unroll
            for (int p = 1; p < 8; ++p) v[s][p] = cmulc(v[s][p], tw.w1[p - 1]);
            rh_xw_write<V, true>(v[s], xbuf, s, w, l_hi, l_lo);
            rh_xw_read<V, true>(v[s], xbuf, s, w, l_hi, l_lo);
        }
#pragma unroll
        for (int s = 0; s < V; ++s) {
            Bfly<8, +1>::run(v[s]);
#pragma unroll
            for (int p = 1; p < 8; ++p) v[s][p] = cmulc(v[s][p], w2[p - 1]);
            rh_xw_write<V, false>(v[s], xbuf, s, w, l_hi, l_lo);
            rh_xw_read<V, false>(v[s], xbuf, s, w, l_hi, l_lo);
        }
#pragma unroll
        for (int s = 0; s < V; ++s) Bfly<8, +1>::run(v[s]);
    } else if (V > 1 && RH_INTERLEAVE) {
#pragma unroll
        for (int s = 0; s < V; ++s) {
            Bfly<8, +1>::run(v[s]);
#pragma unroll
            for (int p = 1; p < 8; ++p) v[s][p] = cmulc(v[s][p], tw.w1[p - 1]);
        }
        rh_xch_wave_all<V, true>(v, xbuf, w, l_hi, l_lo);
        cf w2[7];
#pragma unroll
        for (int p = 1; p < 8; ++p) w2[p - 1] = lds_rd(&tw.w2[p * 8 + l_lo]);
#pragma unroll
        for (int s = 0; s < V; ++s) {
            Bfly<8, +1>::run(v[s]);
#pragma unroll
            for (int p = 1; p < 8; ++p) v[s][p] = cmulc(v[s][p], w2[p - 1]);
        }
        rh_xch_wave_all<V, false>(v, xbuf, w, l_hi, l_lo);
#pragma unroll
        for (int s = 0; s < V; ++s) Bfly<8, +1>::run(v[s]);
    } else {
#pragma unroll
    for (int s = 0; s < V; ++s) {
        cf *slice = xbuf + s * RowH<V>::XSUB + w * Row8::SLICE;
        Bfly<8, +1>::run(v[s]);
#pragma unroll
        for (int p = 1; p < 8; ++p) v[s][p] = cmulc(v[s][p], tw.w1[p - 1]);
        r8_xch_wave<true>(v[s], slice, l_hi, l_lo);
        Bfly<8, +1>::run(v[s]);
#pragma unroll
        for (int p = 1; p < 8; ++p) v[s][p] = cmulc(v[s][p], lds_rd(&tw.w2[p * 8 + l_lo]));
        r8_xch_wave<false>(v[s], slice, l_hi, l_lo);
        Bfly<8, +1>::run(v[s]);
    }
    }
    if (V == 2) {
        cf wj[8];                                     // W^{-(jb + 512 e)} = conj(wq) * exp(+2 pi i e/16)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            cf f1 = cmulc(v[V - 1][e], wq);
            switch (e) {                              // multiply by the conjugate 16th root (backward sign)
            case 1: f1 = mul_w16<1, +1>(f1); break; case 2: f1 = mul_w16<2, +1>(f1); break; case 3: f1 = mul_w16<3, +1>(f1); break;
            case 4: f1 = mul_w16<4, +1>(f1); break; case 5: f1 = mul_w16<5, +1>(f1); break; case 6: f1 = mul_w16<6, +1>(f1); break;
            case 7: f1 = mul_w16<7, +1>(f1); break; default: break;
            }
            wj[e] = f1;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { const cf f0 = v[0][e]; v[0][e] = cadd(f0, wj[e]); v[V - 1][e] = csub(f0, wj[e]); }
    }
}

// forward: the transposed sequence: w[j'] in v[0][e], w[j' + 4096] in v[1][e] (digit-reversed) -> W[V q + s] in v[s][e], q = t + 512 e
template <int V> FB_DEV void rh_fwd(cf (*v)[8], cf *xbuf, const Row8Tw &tw, cf wq, int w, int l)
{
    const int l_hi = l >> 3, l_lo = l & 7;
    if (V == 2) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const cf a = v[0][e], b = v[V - 1][e];
            v[0][e] = cadd(a, b);
            cf d = cmul(csub(a, b), wq);              // (a - b) W^{jb + 512 e}
            switch (e) {
            case 1: d = mul_w16<1, -1>(d); break; case 2: d = mul_w16<2, -1>(d); break; case 3: d = mul_w16<3, -1>(d); break;
            case 4: d = mul_w16<4, -1>(d); break; case 5: d = mul_w16<5, -1>(d); break; case 6: d = mul_w16<6, -1>(d); break;
            case 7: d = mul_w16<7, -1>(d); break; default: break;
            }
            v[V - 1][e] = d;
        }
    }
    if (V > 1 && RH_INTERLEAVE == 2) {
        cf w2[7];
#pragma unroll
        for (int p = 1; p < 8; ++p) w2[p - 1] = lds_rd(&tw.w2[p * 8 + l_lo]);
#pragma unroll
        for (int s = 0; s < V; ++s) {
            Bfly<8, -1>::run(v[s]);
            rh_xw_write<V, false>(v[s], xbuf, s, w, l_hi, l_lo);
            rh_xw_read<V, false>(v[s], xbuf, s, w, l_hi, l_lo);
        }
#pragma unroll
        for (int s = 0; s < V; ++s) {
#pragma unroll
            for (int p = 1; p < 8; ++p) v[s][p] = cmul(v[s][p], w2[p - 1]);
            Bfly<8, -1>::run(v[s]);
            rh_xw_write<V, true>(v[s], xbuf, s, w, l_hi, l_lo);
            rh_xw_read<V, true>(v[s], xbuf, s, w, l_hi, l_lo);
        }
#pragma unroll
        for (int s = 0; s < V; ++s) {
#pragma unroll
            for (int p = 1; p < 8; ++p) v[s][p] = cmul(v[s][p], tw.w1[p - 1]);
            Bfly<8, -1>::run(v[s]);
        }
    } else if (V > 1 && RH_INTERLEAVE) {
#pragma unroll
        for (int s = 0; s < V; ++s) Bfly<8, -1>::run(v[s]);
        rh_xch_wave_all<V, false>(v, xbuf, w, l_hi, l_lo);
        cf w2[7];
#pragma unroll
        for (int p = 1; p < 8; ++p) w2[p - 1] = lds_rd(&tw.w2[p * 8 + l_lo]);
#pragma unroll
        for (int s = 0; s < V; ++s) {
#pragma unroll
            for (int p = 1; p < 8; ++p) v[s][p] = cmul(v[s][p], w2[p - 1]);
            Bfly<8, -1>::run(v[s]);
        }
        rh_xch_wave_all<V, true>(v, xbuf, w, l_hi, l_lo);
#pragma unroll
        for (int s = 0; s < V; ++s) {
#pragma unroll
            for (int p = 1; p < 8; ++p) v[s][p] = cmul(v[s][p], tw.w1[p - 1]);
            Bfly<8, -1>::run(v[s]);
        }
    } else {
#pragma unroll
    for (int s = 0; s < V; ++s) {
        cf *slice = xbuf + s * RowH<V>::XSUB + w * Row8::SLICE;
        Bfly<8, -1>::run(v[s]);
        r8_xch_wave<false>(v[s], slice, l_hi, l_lo);
#pragma unroll
        for (int p = 1; p < 8; ++p) v[s][p] = cmul(v[s][p], lds_rd(&tw.w2[p * 8 + l_lo]));
        Bfly<8, -1>::run(v[s]);
        r8_xch_wave<true>(v[s], slice, l_hi, l_lo);
#pragma unroll
        for (int p = 1; p < 8; ++p) v[s][p] = cmul(v[s][p], tw.w1[p - 1]);
        Bfly<8, -1>::run(v[s]);
    }
    }
    rh_xch_group<V>(v, xbuf, w, l);
#pragma unroll
    for (int s = 0; s < V; ++s) {
#pragma unroll
        for (int p = 1; p < 8; ++p) v[s][p] = cmul(v[s][p], tw.w0[p - 1]);
        Bfly<8, -1>::run(v[s]);
    }
}

// ---- staging: LDS-DMA of one half-spectrum row X[0..M] ---------------------------------------------------------------
// LDS address of a generic pointer into LDS = its low 32 bits (the aperture lives in the high half).  Going through the
// integer avoids the generic -> LDS address-space cast, whose null check trips a backend assertion in this kernel when
// scalar registers run short ("Illegal instruction detected: V_CMP_NE_U32_e32 0, $src_shared_base").
typedef void __attribute__((address_space(3))) *rh_lds_ptr;
FB_DEV rh_lds_ptr rh_to_lds(const void *p) { return (rh_lds_ptr)(unsigned)(size_t)p; }
template <int V, bool SLAB>
FB_DEV void rh_dma_issue(cf *stg, int t, const RowView &view, int field, int row)
{
    constexpr int M = RowH<V>::M;
    const int w = t >> 6, lane = t & 63;
#pragma unroll
    for (int c = 0; c < 4 * V; ++c) {                 // 32 V chunks of 1 KiB, 4 V per wave
        const int ch = w + c * 8, k = ch * 128 + lane * 2;
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)row_ptr<SLAB>(view, field, row, k),
                                         rh_to_lds(stg + ch * 128), 16, 0, RH_NT ? 2 : 0);
    }
    // X[M]: one dword per lane (lanes 0, 1)
    rh_lds_ptr nyq = rh_to_lds(stg + M);
    const float *src = reinterpret_cast<const float *>(row_ptr<SLAB>(view, field, row, M)) + (t & 1);
    if (t < 2) __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)src, nyq, 4, 0, 0);
}
#define RH_WAIT_ROW() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

// c2r pre-processing from the staged row into the first backward stage's registers
template <int V> FB_DEV void rh_ext(cf (*v)[8], int t, const cf *stg, const cf *wx /*[V]: exp(+2 pi i (V t + s)/N)*/)
{
    constexpr int M = RowH<V>::M;
#ifdef FB_R8_NOEXT  /* timing experiment only (results are wrong): no staging reads, no pre-processing */
#pragma unroll
    for (int s = 0; s < V; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) v[s][e] = cf_make(wx[s].x + (float)e, wx[s].y);
    return;
#endif
#pragma unroll
    for (int s = 0; s < V; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = V * (t + 512 * e) + s;
            cf a = lds_rd(&stg[k]), b = lds_rd(&stg[M - k]);
            if (e == 0 && s == 0 && t == 0) { a.y = 0.f; b.y = 0.f; }                  // k = 0: X[0] and X[M] count as real
            const cf ev = cadd_conj(a, b);                               // X[k] + conj X[M-k]
            cf d = cmul(csub_conj(a, b), wx[s]);                         // (X[k] - conj X[M-k]) e^{2 pi i (V t + s)/N} ...
            switch (e) {                                                               // ... e^{2 pi i e/16}
            case 1: d = mul_w16<1, +1>(d); break; case 2: d = mul_w16<2, +1>(d); break; case 3: d = mul_w16<3, +1>(d); break;
            case 4: d = mul_w16<4, +1>(d); break; case 5: d = mul_w16<5, +1>(d); break; case 6: d = mul_w16<6, +1>(d); break;
            case 7: d = mul_w16<7, +1>(d); break; default: break;
            }
            v[s][e] = cadd_ib(ev, d);                                                  // E + i O
        }
}

// vort_src [x][y] -> the kernel's physical-space order: float2 (y = 2j, 2j+1) at [x][s'][e][t], j = jb(t) + 512 e + 4096 s'
template <int V>
__global__ void __launch_bounds__(256) k_rowh_permute_src(const float *__restrict__ in, float *__restrict__ out, int nrows)
{
    constexpr int M = RowH<V>::M;
    const size_t total = (size_t)nrows * M;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(idx / M), r = (int)(idx - (size_t)x * M);
        const int t = r & 511, e = (r >> 9) & 7, sp = r >> 12;
        const int w = t >> 6, l = t & 63, jb = w + 8 * (l >> 3) + 64 * (l & 7);
        const int j = jb + 512 * e + 4096 * sp;
        reinterpret_cast<float2 *>(out)[idx] = reinterpret_cast<const float2 *>(in)[(size_t)x * M + j];
    }
}

template <int V, bool SLAB>
__global__ void __launch_bounds__(512, RowH<V>::MIN_WAVES) k_rowh(RowArgs a, const cf *__restrict__ root4096 /* W_4096^j */, const cf *__restrict__ rootN /* W_N^j, N = 8192 V */)
{
    using C = RowH<V>;
    constexpr int M = C::M, N = C::N;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf *xbuf = reinterpret_cast<cf *>(smem_raw);
    cf *stg = xbuf + C::XBUF;
    const int t = threadIdx.x, w = t >> 6, l = t & 63;
    Row8Tw tw;
#pragma unroll
    for (int p = 1; p < 8; ++p) { tw.w0[p - 1] = root4096[p * t]; tw.w1[p - 1] = root4096[8 * p * l]; }
    cf *tw2 = stg + C::STG;
    if (t < 64) tw2[t] = root4096[64 * (t & 7) * (t >> 3)];          // [p = t >> 3][l_lo = t & 7]
    tw.w2 = tw2;
    const int jb = w + 8 * (l >> 3) + 64 * (l & 7);                   // digit-reversed base index of this thread's outputs
    const cf wq = V == 2 ? rootN[2 * jb] : cf_make(1.f, 0.f);         // W_M^{jb} = W_N^{2 jb} (forward sign)
    cf wx[V];                                                        // exp(+2 pi i (V t + s)/N) = conj(W_N^{V t + s})
#pragma unroll
    for (int s = 0; s < V; ++s) { const cf r = rootN[V * t + s]; wx[s] = cf_make(r.x, -r.y); }
#pragma unroll
    for (int p = 0; p < 7; ++p) asm volatile("" :: "v"(tw.w0[p]), "v"(tw.w1[p]));      // land the table loads here, not behind a prefetch
    asm volatile("" :: "v"(wq), "v"(wx[0]));
    __syncthreads();

    const int iters = (a.nx + gridDim.x - 1) / gridDim.x;
#ifdef FB_ROW_SAMEROW   /* timing experiment only: every workgroup works on row 0 (no HBM traffic); results are wrong */
    auto row_of = [&](int it, bool &valid) { const int r = it * gridDim.x + blockIdx.x; valid = r < a.nx; return a.x0; };
#else
    auto row_of = [&](int it, bool &valid) { const int r = it * gridDim.x + blockIdx.x; valid = r < a.nx; return a.x0 + (valid ? r : 0); };
#endif
    if (iters > 0) {
        bool vld; const int x = row_of(0, vld);
        rh_dma_issue<V, SLAB>(stg, t, a.M, 0, x);
        RH_WAIT_ROW();
    }
    // field order of the four backward transforms: d vort/dx (0), d psi/dy (2), d vort/dy (1), d psi/dx (3)
    for (int it = 0; it < iters; ++it) {
        bool valid;
        const int x = row_of(it, valid);                              // an invalid workgroup recomputes row 0, stores nothing
        bool vn = false;
        const int xn = (it + 1 < iters) ? row_of(it + 1, vn) : -1;
        cf v[V][8];
        cf p[V][8];                                                   // dvortdx, then -u * dvortdx (pairs y = 2j, 2j+1)
        // one backward transform: wait for its row, pre-process it out of the staging area, send for the next row, transform.
        // Per-phase opaque thread id: otherwise the staging and exchange addresses of all four phases stay live in registers.
        auto c2r_phase = [&](bool wait, int next_field, int next_row) {
            const int tp = RH_LAUNDER(t);
            if (wait) RH_WAIT_ROW();                                  // (the first phase's row was waited for before the previous stores)
            lds_barrier();
            rh_ext<V>(v, tp, stg, wx);
#ifndef RH_LATE_DMA   /* -DRH_LATE_DMA: the next row is sent for behind the exchange barrier instead (one barrier fewer per phase; measured: no gain at ny = 4096 and 16384, 2 % slower in k_rowh2) */
            lds_barrier();
            if (next_row >= 0) rh_dma_issue<V, SLAB>(stg, tp, a.M, next_field, next_row);
            rh_bwd<V>(v, xbuf, tw, wq, tp >> 6, tp & 63, RhNothing());
#else
            // the next row is sent for behind the barrier of the transform's own workgroup-wide exchange: by then every wave has
            // read what it needs of the staged row
            rh_bwd<V>(v, xbuf, tw, wq, tp >> 6, tp & 63, [&] { if (next_row >= 0) rh_dma_issue<V, SLAB>(stg, tp, a.M, next_field, next_row); });
#endif
        };
        c2r_phase(false, 2, x);                                       // d vort/dx                         main.cpp:154
#pragma unroll
        for (int s = 0; s < V; ++s)
#pragma unroll
            for (int e = 0; e < 8; ++e) p[s][e] = cf_make(v[s][e].x * a.scale, v[s][e].y * a.scale);
        c2r_phase(true, 1, x);                                        // d psi/dy: u = -(c2r * scale)  =>  -u * dvortdx = (c2r * scale) * dvortdx   main.cpp:200-201,225
#pragma unroll
        for (int s = 0; s < V; ++s)
#pragma unroll
            for (int e = 0; e < 8; ++e) p[s][e] = cf_make((v[s][e].x * a.scale) * p[s][e].x, (v[s][e].y * a.scale) * p[s][e].y);
        c2r_phase(true, 3, x);                                        // d vort/dy                         main.cpp:168
        {
            cf zy[V][8];
#pragma unroll
            for (int s = 0; s < V; ++s)
#pragma unroll
                for (int e = 0; e < 8; ++e) zy[s][e] = cf_make(v[s][e].x * a.scale, v[s][e].y * a.scale);
            c2r_phase(true, 0, xn);                                   // d psi/dx = v; the next x row's first field travels meanwhile
#pragma unroll
            for (int s = 0; s < V; ++s)
#pragma unroll
                for (int e = 0; e < 8; ++e)                           // - u*dvortdx - v*dvortdy          main.cpp:214,225-227
                    v[s][e] = cf_make(p[s][e].x - (v[s][e].x * a.scale) * zy[s][e].x, p[s][e].y - (v[s][e].y * a.scale) * zy[s][e].y);
        }
        const int tt = RH_LAUNDER(t), wl = tt >> 6, ll = tt & 63;
        if (a.src && a.src_nz[x]) {                                   // ... + vort_src (permuted order); the loads and their wait stay in this branch; rows of zeros are skipped
            const float2 *sp = reinterpret_cast<const float2 *>(a.src) + (size_t)x * M + tt;
#pragma unroll
            for (int s = 0; s < V; ++s)
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float2 q = sp[(s * 8 + e) * 512]; v[s][e].x += q.x; v[s][e].y += q.y; }
        }
        rh_fwd<V>(v, xbuf, tw, wq, wl, ll);                           // main.cpp:237 (y part)
        RH_WAIT_ROW();                                                // the next row's first field has landed
        // r2c post-processing: the upper half of W (k >= M/2: e >= 4) goes through LDS at position k - M/2
        lds_barrier();
#pragma unroll
        for (int s = 0; s < V; ++s)
#pragma unroll
            for (int e = 4; e < 8; ++e) lds_wr(&xbuf[V * (tt + 512 * (e - 4)) + s], v[s][e]);
        lds_barrier();
#ifdef FB_R8_NOST   /* timing experiment only: (almost) no stores */
        if (valid && v[0][0].x == 123.456f) {
#else
        if (valid) {
#endif
#pragma unroll
            for (int s = 0; s < V; ++s)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = V * (tt + 512 * e) + s;             // 0 <= k < M/2
                    const cf wk = v[s][e];
                    if (k == 0) {                                     // T[0] = Re W0 + Im W0 ; T[M] = Re W0 - Im W0   (both real)
                        if (row_keep<SLAB>(a.T, a.t_frozen, 0)) *const_cast<cf *>(row_ptr<SLAB>(a.T, 0, x, 0)) = cf_make(wk.x + wk.y, 0.f);
                        if (row_keep<SLAB>(a.T, a.t_frozen, M)) *const_cast<cf *>(row_ptr<SLAB>(a.T, 0, x, M)) = cf_make(wk.x - wk.y, 0.f);
                        continue;
                    }
                    const cf wm = lds_rd(&xbuf[M / 2 - k]);           // W[M - k] sits at (M - k) - M/2
                    const cf ev = cf_make(0.5f * (wk.x + wm.x), 0.5f * (wk.y - wm.y));          // E = (W[k] + conj W[M-k]) / 2
                    const cf od = cf_make(0.5f * (wk.y + wm.y), 0.5f * (wm.x - wk.x));          // O = (W[k] - conj W[M-k]) / (2i)
                    cf co = cmulc(od, wx[s]);                                                    // e^{-2 pi i (V t + s)/N} O ...
                    switch (e) {                                                                 // ... e^{-2 pi i e/16}
                    case 1: co = mul_w16<1, -1>(co); break; case 2: co = mul_w16<2, -1>(co); break; case 3: co = mul_w16<3, -1>(co); break;
                    default: break;
                    }
                    if (row_keep<SLAB>(a.T, a.t_frozen, k)) st2<false>(const_cast<cf *>(row_ptr<SLAB>(a.T, 0, x, k)), cadd(ev, co));
                    const cf tm = csub(ev, co);
                    if (row_keep<SLAB>(a.T, a.t_frozen, M - k)) st2<false>(const_cast<cf *>(row_ptr<SLAB>(a.T, 0, x, M - k)), cf_make(tm.x, -tm.y));
                }
            if (tt == 0 && row_keep<SLAB>(a.T, a.t_frozen, M / 2)) {                             // k = M/2 is its own mirror: T = conj W
                const cf wh = v[0][4];
                *const_cast<cf *>(row_ptr<SLAB>(a.T, 0, x, M / 2)) = cf_make(wh.x, -wh.y);
            }
        }
    }
}


// =====================================================================================================================
// k_rowh2: ny = 8192 row pass with the last radix-2 step of the x transform fused in (nx = 8192 = 2 x 4096, fb_col_full.h).
// A 1024-thread workgroup owns one x2: half h (waves 8h .. 8h+7) produces the physical row x = x2 + 4096 h,
//     X_h[k] = Y_0[x2][k] + (-1)^h e^{+2 pi i x2/8192} Y_1[x2][k]           (both staged rows are read by both halves)
// runs k_rowh<1>'s five transforms on it, and the halves meet again after the r2c post-processing:
//     U_0[x2] = T_0 + T_1,   U_1[x2] = e^{-2 pi i x2/8192} (T_0 - T_1)      (half h stores U_h at row 4096 h + x2)
// Same LDS budget as k_rowh<2>: two exchange buffers + two staged rows = 140 KB, one workgroup (16 waves) per CU.
// =====================================================================================================================
struct RowH2 {
    static constexpr int M = 4096, XSUB = RowH<1>::XSUB, STG = RowH<1>::STG, TW2 = 64;
    static constexpr size_t LDS_BYTES = (size_t)(2 * XSUB + 2 * STG + TW2) * sizeof(cf);
};

FB_DEV void rh2_ext(cf *v, int t, const cf *stg0, const cf *stg1, cf wh, cf wx)
{
    constexpr int M = RowH2::M;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = t + 512 * e;
        cf a = cadd(lds_rd(&stg0[k]), cmul(lds_rd(&stg1[k]), wh));                    // X_h[k]
        cf b = cadd(lds_rd(&stg0[M - k]), cmul(lds_rd(&stg1[M - k]), wh));            // X_h[M-k]
        if (e == 0 && t == 0) { a.y = 0.f; b.y = 0.f; }                               // k = 0: X[0] and X[M] count as real
        const cf ev = cadd_conj(a, b);
        cf d = cmul(csub_conj(a, b), wx);
        switch (e) {
        case 1: d = mul_w16<1, +1>(d); break; case 2: d = mul_w16<2, +1>(d); break; case 3: d = mul_w16<3, +1>(d); break;
        case 4: d = mul_w16<4, +1>(d); break; case 5: d = mul_w16<5, +1>(d); break; case 6: d = mul_w16<6, +1>(d); break;
        case 7: d = mul_w16<7, +1>(d); break; default: break;
        }
        v[e] = cadd_ib(ev, d);
    }
}

__global__ void __launch_bounds__(1024) k_rowh2(RowArgs a, const cf *__restrict__ root4096, const cf *__restrict__ rootN /* W_8192^j (y) */)
{
    constexpr int M = RowH2::M;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf *smem = reinterpret_cast<cf *>(smem_raw);
    const int h = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 9));            // half: wave-uniform
    const int t = threadIdx.x & 511, w = t >> 6, l = t & 63;
    cf *xbuf = smem + h * RowH2::XSUB, *xoth = smem + (1 - h) * RowH2::XSUB;
    cf *stg0 = smem + 2 * RowH2::XSUB, *stg1 = stg0 + RowH2::STG, *stgh = h ? stg1 : stg0;
    Row8Tw tw;
#pragma unroll
    for (int p = 1; p < 8; ++p) { tw.w0[p - 1] = root4096[p * t]; tw.w1[p - 1] = root4096[8 * p * l]; }
    cf *tw2 = stg1 + RowH2::STG;
    if (threadIdx.x < 64) tw2[t] = root4096[64 * (t & 7) * (t >> 3)];
    tw.w2 = tw2;
    const cf wq = cf_make(1.f, 0.f);
    cf wx;                                                           // exp(+2 pi i t/N), N = 8192
    { const cf r = rootN[t]; wx = cf_make(r.x, -r.y); }
#pragma unroll
    for (int p = 0; p < 7; ++p) asm volatile("" :: "v"(tw.w0[p]), "v"(tw.w1[p]));
    asm volatile("" :: "v"(wx));
    __syncthreads();

    const long sub = a.sub_rows;
    const int iters = (a.nx + gridDim.x - 1) / gridDim.x;
    auto row_of = [&](int it, bool &valid) { const int r = it * gridDim.x + blockIdx.x; valid = r < a.nx; return a.x0 + (valid ? r : 0); };
    if (iters > 0) {
        bool vld; const int x2 = row_of(0, vld);
        rh_dma_issue<1, false>(stgh, t, a.M, 0, (int)(h * sub) + x2);
        RH_WAIT_ROW();
    }
    for (int it = 0; it < iters; ++it) {
        bool valid;
        const int x2 = row_of(it, valid);
        bool vn = false;
        const int xn = (it + 1 < iters) ? row_of(it + 1, vn) : -1;
        const cf wb = a.tw_x[x2];                                     // W_nx^{x2} (forward sign), wave-uniform
        const cf wh = h ? cf_make(-wb.x, wb.y) : cf_make(wb.x, -wb.y);   // (-1)^h e^{+2 pi i x2/nx}
        const int xrow = (int)(h * sub) + x2;                         // physical row x2 + 4096 h == storage row of Y_h / U_h
        cf v1[1][8];
        cf (&v)[8] = v1[0];
        cf p[8];
        auto c2r_phase = [&](bool wait, int next_field, int next_x2) {
            const int tp = launder(t);
            if (wait) RH_WAIT_ROW();
            lds_barrier();
            rh2_ext(v, tp, stg0, stg1, wh, wx);
#ifndef RH_LATE_DMA
            lds_barrier();
            if (next_x2 >= 0) rh_dma_issue<1, false>(stgh, tp, a.M, next_field, (int)(h * sub) + next_x2);
            rh_bwd<1>(v1, xbuf, tw, wq, tp >> 6, tp & 63, RhNothing());
#else
            rh_bwd<1>(v1, xbuf, tw, wq, tp >> 6, tp & 63, [&] { if (next_x2 >= 0) rh_dma_issue<1, false>(stgh, tp, a.M, next_field, (int)(h * sub) + next_x2); });
#endif
        };
        c2r_phase(false, 2, x2);                                      // d vort/dx                         main.cpp:154
#pragma unroll
        for (int e = 0; e < 8; ++e) p[e] = cf_make(v[e].x * a.scale, v[e].y * a.scale);
        c2r_phase(true, 1, x2);                                       // d psi/dy                          main.cpp:200-201,225
#pragma unroll
        for (int e = 0; e < 8; ++e) p[e] = cf_make((v[e].x * a.scale) * p[e].x, (v[e].y * a.scale) * p[e].y);
        c2r_phase(true, 3, x2);                                       // d vort/dy                         main.cpp:168
        {
            cf zy[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) zy[e] = cf_make(v[e].x * a.scale, v[e].y * a.scale);
            c2r_phase(true, 0, xn);                                   // d psi/dx                          main.cpp:214,225-227
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = cf_make(p[e].x - (v[e].x * a.scale) * zy[e].x, p[e].y - (v[e].y * a.scale) * zy[e].y);
        }
        const int tt = launder(t), wl = tt >> 6, ll = tt & 63;
        if (a.src && a.src_nz[xrow]) {
            const float2 *sp = reinterpret_cast<const float2 *>(a.src) + (size_t)xrow * M + tt;
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float2 q = sp[e * 512]; v[e].x += q.x; v[e].y += q.y; }
        }
        rh_fwd<1>(v1, xbuf, tw, wq, wl, ll);                          // main.cpp:237 (y part)
        RH_WAIT_ROW();
        lds_barrier();
#pragma unroll
        for (int e = 4; e < 8; ++e) lds_wr(&xbuf[tt + 512 * (e - 4)], v[e]);       // W[k], k >= M/2, at k - M/2
        lds_barrier();
        cf tk[4], tm[4];                                              // T_h[k], T_h[M-k], k = t + 512 e < M/2
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = tt + 512 * e;
            const cf wk = v[e];
            if (e == 0 && tt == 0) { tk[0] = cf_make(wk.x + wk.y, 0.f); tm[0] = cf_make(wk.x - wk.y, 0.f); continue; }   // T[0], T[M]
            const cf wm = lds_rd(&xbuf[M / 2 - k]);
            const cf ev = cf_make(0.5f * (wk.x + wm.x), 0.5f * (wk.y - wm.y));
            const cf od = cf_make(0.5f * (wk.y + wm.y), 0.5f * (wm.x - wk.x));
            cf co = cmulc(od, wx);
            switch (e) {
            case 1: co = mul_w16<1, -1>(co); break; case 2: co = mul_w16<2, -1>(co); break; case 3: co = mul_w16<3, -1>(co); break;
            default: break;
            }
            tk[e] = cadd(ev, co);
            const cf d = csub(ev, co);
            tm[e] = cf_make(d.x, -d.y);
        }
        const cf th = cf_make(v[4].x, -v[4].y);                       // thread 0: T[M/2] = conj W[M/2]
        // the halves swap their spectra through the exchange buffers (natural index, M + 1 slots)
        lds_barrier();
#pragma unroll
        for (int e = 0; e < 4; ++e) { const int k = tt + 512 * e; lds_wr(&xbuf[k], tk[e]); lds_wr(&xbuf[M - k], tm[e]); }
        if (tt == 0) lds_wr(&xbuf[M / 2], th);
        lds_barrier();
        if (valid) {
            auto comb = [&](cf mine, cf other) { return h ? cmul(csub(other, mine), wb) : cadd(mine, other); };   // U_0 = T_0 + T_1 ; U_1 = W_nx^{x2} (T_0 - T_1)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = tt + 512 * e;
                st2<false>(const_cast<cf *>(row_ptr<false>(a.T, 0, xrow, k)), comb(tk[e], lds_rd(&xoth[k])));
                st2<false>(const_cast<cf *>(row_ptr<false>(a.T, 0, xrow, M - k)), comb(tm[e], lds_rd(&xoth[M - k])));
            }
            if (tt == 0) *const_cast<cf *>(row_ptr<false>(a.T, 0, xrow, M / 2)) = comb(th, lds_rd(&xoth[M / 2]));
        }
    }
}


#ifdef RH2_SPLIT_EXPERIMENT
// =====================================================================================================================
// k_rowh2s -- TIMING EXPERIMENT of round 4 (never compiled into the product: results are wrong, the forward
// radix-2 step over x is left out).  k_rowh2 as TWO independent 512-thread workgroups per x2, two resident per CU as k_rowh<1>:
// workgroup (x2, h) produces the physical row x = x2 + 4096 h alone.  It needs both half-transformed rows Y_0[x2], Y_1[x2] of every
// field, and a CU's LDS holds one exchange buffer (37 KB) + ONE staged row (32 KB) per context when two contexts share it (two staged
// rows: 101 KB, one context per CU).  So Y_0 arrives by LDS-DMA one phase ahead as before and Y_1 is read into registers at the
// point of use (8 B per lane, 512 B per wave instruction); the partner workgroup (h' = 1 - h) is the block 8 ids further on, i.e. the
// next one on the same XCD, so that the second reader of a row can find it in that XCD's L2.  The tendency row T_h is stored as it
// is: in the real variant k_col_full<., 2> would form U_k1 = W^{k1 x2}(T_0 + (-1)^{k1} T_1) on load (a second read of the tendency).
// =====================================================================================================================
#ifndef RH2S_HALVES
#define RH2S_HALVES 1      /* 1: Y_1 in two batches of four e (16 registers in flight, the second batch travels while the first is used); 0: all sixteen loads at once */
#endif
FB_DEV void rh2s_one(cf *v, int e, int t, const cf *stg0, cf y1k, cf y1m, cf wh, cf wx)
{
    constexpr int M = RowH2::M;
    const int k = t + 512 * e;
    cf a = cadd(lds_rd(&stg0[k]), cmul(y1k, wh));                                     // X_h[k]
    cf b = cadd(lds_rd(&stg0[M - k]), cmul(y1m, wh));                                 // X_h[M-k]
    if (e == 0 && t == 0) { a.y = 0.f; b.y = 0.f; }
    const cf ev = cadd_conj(a, b);
    cf d = cmul(csub_conj(a, b), wx);
    switch (e) {
    case 1: d = mul_w16<1, +1>(d); break; case 2: d = mul_w16<2, +1>(d); break; case 3: d = mul_w16<3, +1>(d); break;
    case 4: d = mul_w16<4, +1>(d); break; case 5: d = mul_w16<5, +1>(d); break; case 6: d = mul_w16<6, +1>(d); break;
    case 7: d = mul_w16<7, +1>(d); break; default: break;
    }
    v[e] = cadd_ib(ev, d);
}
FB_DEV void rh2s_ext(cf *v, int t, const cf *stg0, const cf *__restrict__ g1, cf wh, cf wx)
{
    constexpr int M = RowH2::M;
#if RH2S_HALVES
    cf ya[4], yb[4], yc[4], yd[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { const int k = t + 512 * e; ya[e] = g1[k]; yb[e] = g1[M - k]; }
    RH_WAIT_ROW();                                                    // Y_0 (sent for a phase ago) and the eight loads above
    lds_barrier();
#pragma unroll
    for (int e = 4; e < 8; ++e) { const int k = t + 512 * e; yc[e - 4] = g1[k]; yd[e - 4] = g1[M - k]; }
#pragma unroll
    for (int e = 0; e < 4; ++e) rh2s_one(v, e, t, stg0, ya[e], yb[e], wh, wx);
    RH_WAIT_ROW();
#pragma unroll
    for (int e = 4; e < 8; ++e) rh2s_one(v, e, t, stg0, yc[e - 4], yd[e - 4], wh, wx);
#else
    cf y1k[8], y1m[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { const int k = t + 512 * e; y1k[e] = g1[k]; y1m[e] = g1[M - k]; }
    RH_WAIT_ROW();                                                    // Y_0 (sent for a phase ago) and the sixteen loads above
    lds_barrier();
#pragma unroll
    for (int e = 0; e < 8; ++e) rh2s_one(v, e, t, stg0, y1k[e], y1m[e], wh, wx);
#endif
}

__global__ void __launch_bounds__(512, 4) k_rowh2s(RowArgs a, const cf *__restrict__ root4096, const cf *__restrict__ rootN, int pairs_per_iter)
{
    using C = RowH<1>;
    constexpr int M = C::M;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf *xbuf = reinterpret_cast<cf *>(smem_raw);
    cf *stg = xbuf + C::XBUF;
    const int t = threadIdx.x, w = t >> 6, l = t & 63;
    Row8Tw tw;
#pragma unroll
    for (int p = 1; p < 8; ++p) tw.w0[p - 1] = root4096[p * t];
    cf *tw2 = stg + C::STG;
    if (t < 64) tw2[t] = root4096[64 * (t & 7) * (t >> 3)];
    tw.w2 = tw2;
    cf *tw1 = tw2 + C::TW2;                                          // W_512^{p l} at [p - 1][l]: read per phase instead of living in 14 registers
    if (t < 448) tw1[t] = root4096[8 * (1 + (t >> 6)) * (t & 63)];
    const cf wq = cf_make(1.f, 0.f);
    cf wx;
    { const cf r = rootN[t]; wx = cf_make(r.x, -r.y); }
#pragma unroll
    for (int p = 0; p < 7; ++p) asm volatile("" :: "v"(tw.w0[p]));
    asm volatile("" :: "v"(wx));
    __syncthreads();

    // blocks b and b + 8 (the same XCD, consecutive there) are the two halves of one x2
    const int b = blockIdx.x, xcd = b & 7, j = b >> 3, h = j & 1, q = (j >> 1) * 8 + xcd;
    const long sub = a.sub_rows;
    const int iters = (a.nx + pairs_per_iter - 1) / pairs_per_iter;
    auto row_of = [&](int it, bool &valid) { const int r = it * pairs_per_iter + q; valid = r < a.nx; return a.x0 + (valid ? r : 0); };
    if (iters > 0) { bool vld; const int x2 = row_of(0, vld); rh_dma_issue<1, false>(stg, t, a.M, 0, x2); }
    for (int it = 0; it < iters; ++it) {
        bool valid;
        const int x2 = row_of(it, valid);
        bool vn = false;
        const int xn = (it + 1 < iters) ? row_of(it + 1, vn) : -1;
        const cf wb = a.tw_x[x2];
        const cf wh = h ? cf_make(-wb.x, wb.y) : cf_make(wb.x, -wb.y);
        const int xrow = (int)(h * sub) + x2;
        cf v1[1][8];
        cf (&v)[8] = v1[0];
        cf p[8];
        auto c2r_phase = [&](int field, int next_field, int next_x2) {
            const int tp = launder(t);
            rh2s_ext(v, tp, stg, row_ptr<false>(a.M, field, (int)sub + x2, 0), wh, wx);
            lds_barrier();
            if (next_x2 >= 0) rh_dma_issue<1, false>(stg, tp, a.M, next_field, next_x2);
#pragma unroll
            for (int p = 0; p < 7; ++p) tw.w1[p] = lds_rd(&tw1[p * 64 + (tp & 63)]);
            rh_bwd<1>(v1, xbuf, tw, wq, tp >> 6, tp & 63, RhNothing());
        };
        c2r_phase(0, 2, x2);
#pragma unroll
        for (int e = 0; e < 8; ++e) p[e] = cf_make(v[e].x * a.scale, v[e].y * a.scale);
        c2r_phase(2, 1, x2);
#pragma unroll
        for (int e = 0; e < 8; ++e) p[e] = cf_make((v[e].x * a.scale) * p[e].x, (v[e].y * a.scale) * p[e].y);
        c2r_phase(1, 3, x2);
        {
            cf zy[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) zy[e] = cf_make(v[e].x * a.scale, v[e].y * a.scale);
            c2r_phase(3, 0, xn);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = cf_make(p[e].x - (v[e].x * a.scale) * zy[e].x, p[e].y - (v[e].y * a.scale) * zy[e].y);
        }
        const int tt = launder(t), wl = tt >> 6, ll = tt & 63;
#pragma unroll
        for (int p = 0; p < 7; ++p) tw.w1[p] = lds_rd(&tw1[p * 64 + ll]);
        rh_fwd<1>(v1, xbuf, tw, wq, wl, ll);
        lds_barrier();
#pragma unroll
        for (int e = 4; e < 8; ++e) lds_wr(&xbuf[tt + 512 * (e - 4)], v[e]);
        lds_barrier();
        if (valid) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = tt + 512 * e;
                const cf wk = v[e];
                if (k == 0) {
                    *const_cast<cf *>(row_ptr<false>(a.T, 0, xrow, 0)) = cf_make(wk.x + wk.y, 0.f);
                    *const_cast<cf *>(row_ptr<false>(a.T, 0, xrow, M)) = cf_make(wk.x - wk.y, 0.f);
                    continue;
                }
                const cf wm = lds_rd(&xbuf[M / 2 - k]);
                const cf ev = cf_make(0.5f * (wk.x + wm.x), 0.5f * (wk.y - wm.y));
                const cf od = cf_make(0.5f * (wk.y + wm.y), 0.5f * (wm.x - wk.x));
                cf co = cmulc(od, wx);
                switch (e) {
                case 1: co = mul_w16<1, -1>(co); break; case 2: co = mul_w16<2, -1>(co); break; case 3: co = mul_w16<3, -1>(co); break;
                default: break;
                }
                st2<false>(const_cast<cf *>(row_ptr<false>(a.T, 0, xrow, k)), cadd(ev, co));
                const cf tm = csub(ev, co);
                st2<false>(const_cast<cf *>(row_ptr<false>(a.T, 0, xrow, M - k)), cf_make(tm.x, -tm.y));
            }
            if (tt == 0) { const cf wh2 = v[4]; *const_cast<cf *>(row_ptr<false>(a.T, 0, xrow, M / 2)) = cf_make(wh2.x, -wh2.y); }
        }
    }
}
#endif  // RH2_SPLIT_EXPERIMENT
