// fb_fft_core.h -- device-side FFT building blocks for gfx950 (wave64, LDS-staged radix butterflies).
//
// No reference counterpart: the reference takes its FFTs from FFTW3f (main.cpp:126-135).
// Conventions are FFTW's: DIR = -1 forward (exp(-2 pi i jk/n)), DIR = +1 backward, unnormalised.
#pragma once
#include <hip/hip_runtime.h>

// A complex number is a native 2-vector: it lives in an aligned VGPR pair and add/sub map to one
// packed instruction (v_pk_add_f32).  Everything that needs a lane swizzle (complex multiply,
// multiplication by +-i folded into an add) is spelled with VOP3P op_sel/neg modifiers below:
// the compiler's own selection spends 4-5 instructions and register moves per complex multiply.
typedef float cf __attribute__((ext_vector_type(2)));

#define FB_DEV __device__ __forceinline__

FB_DEV cf cf_make(float x, float y) { cf r = {x, y}; return r; }

// LDS read of one complex through a volatile LDS pointer: otherwise the compiler pairs neighbouring reads into
// ds_read2(st64)_b64, which the LDS serves at half the rate of two ds_read_b64 (MI355X_MICROARCH.md, LDS
// table: 8 cycles against 2 + 2).  Measured on k_row8: 0.0867 -> 0.0832 ms per launch.
typedef const volatile __attribute__((address_space(3))) cf *lds_vcf_ptr;
typedef volatile __attribute__((address_space(3))) cf *lds_vcf_wptr;
#ifndef FB_PAIRED_LDS_WRITES   /* ds_write2_b64: 13 cycles against 6 + 6 -- a small but repeatable gain (1080 -> 1094 steps/s) */
FB_DEV void lds_wr(cf *p, cf v) { *(lds_vcf_wptr)p = v; }
#else
FB_DEV void lds_wr(cf *p, cf v) { *p = v; }
#endif
#ifndef FB_PAIRED_LDS_READS
FB_DEV cf lds_rd(const cf *p) { return *(lds_vcf_ptr)p; }
#else
FB_DEV cf lds_rd(const cf *p) { return *p; }
#endif

FB_DEV cf cadd(cf a, cf b) { return a + b; }
FB_DEV cf csub(cf a, cf b) { return a - b; }
#if defined(__HIP_DEVICE_COMPILE__)
// Both instructions of a complex multiply sit in ONE asm statement: the compiler must assume that an asm statement
// it cannot look into writes its result with a destination select and pads every dependent pair of statements with
// s_nop 0 (gfx940 dst-sel forwarding rule; v_pk_*_f32 has no such hazard).  204 of k_rowq's 2492 instructions were
// such padding.  The early-clobber result keeps a and b alive for the second instruction.
#define FB_CMUL_ASM_FUSED(MODS, BC) \
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]\n\tv_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1] " MODS \
        : "=&v"(r) : "v"(a), BC(b))
#define FB_CMUL_ASM_SPLIT(MODS, BC) \
    { cf t_; asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t_) : "v"(a), BC(b)); \
      asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1] " MODS : "=v"(r) : "v"(a), BC(b), "v"(t_)); }
#ifndef FB_SPLIT_CMUL_ASM
#define FB_CMUL_ASM(MODS, BC) FB_CMUL_ASM_FUSED(MODS, BC)
#else
#define FB_CMUL_ASM(MODS, BC) FB_CMUL_ASM_SPLIT(MODS, BC)
#endif
// a*b = (a.x b.x - a.y b.y, a.x b.y + a.y b.x):  t = (a.y b.y, a.y b.x);  r = (a.x b.x - t.x, a.x b.y + t.y)
FB_DEV cf cmul(cf a, cf b)
{
    cf r;
    FB_CMUL_ASM("neg_lo:[0,0,1]", "v");
    return r;
}
// a*conj(b) = (a.x b.x + a.y b.y, a.y b.x - a.x b.y)
FB_DEV cf cmulc(cf a, cf b)
{
    cf r;
    FB_CMUL_ASM("neg_hi:[0,1,0]", "v");
    return r;
}
// same with the second factor in an SGPR pair (compile-time constants)
FB_DEV cf cmul_k(cf a, cf k)
{
    const cf b = k;
    cf r;
    FB_CMUL_ASM("neg_lo:[0,0,1]", "s");
    return r;
}
// the two-statement forms (k_col_full schedules better with them: 0.150 against 0.154 ms at 4096^2, 0.66 against 0.69 ms at 8192^2)
FB_DEV cf cmul_split(cf a, cf b) { cf r; FB_CMUL_ASM_SPLIT("neg_lo:[0,0,1]", "v"); return r; }
FB_DEV cf cmulc_split(cf a, cf b) { cf r; FB_CMUL_ASM_SPLIT("neg_hi:[0,1,0]", "v"); return r; }
FB_DEV cf cmul_k_split(cf a, cf k) { const cf b = k; cf r; FB_CMUL_ASM_SPLIT("neg_lo:[0,0,1]", "s"); return r; }
// a + conj(b) and a - conj(b) in one instruction each (the component-wise form costs two v_pk_add and three v_mov)
FB_DEV cf cadd_conj(cf a, cf b) { cf r; asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
FB_DEV cf csub_conj(cf a, cf b) { cf r; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
// a + i b = (a.x - b.y, a.y + b.x) ;  a - i b = (a.x + b.y, a.y - b.x)
#ifndef FB_SCALAR_ROT
FB_DEV cf cadd_ib(cf a, cf b)
{
    cf r; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r;
}
FB_DEV cf csub_ib(cf a, cf b)
{
    cf r; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r;
}
#else
FB_DEV cf cadd_ib(cf a, cf b) { return cf_make(a.x - b.y, a.y + b.x); }
FB_DEV cf csub_ib(cf a, cf b) { return cf_make(a.x + b.y, a.y - b.x); }
#endif
#else
FB_DEV cf cmul(cf a, cf b) { return cf_make(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
FB_DEV cf cmulc(cf a, cf b) { return cf_make(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }
FB_DEV cf cmul_k(cf a, cf k) { return cmul(a, k); }
FB_DEV cf cmul_split(cf a, cf b) { return cmul(a, b); }
FB_DEV cf cmulc_split(cf a, cf b) { return cmulc(a, b); }
FB_DEV cf cmul_k_split(cf a, cf k) { return cmul(a, k); }
FB_DEV cf cadd_conj(cf a, cf b) { return cf_make(a.x + b.x, a.y - b.y); }
FB_DEV cf csub_conj(cf a, cf b) { return cf_make(a.x - b.x, a.y + b.y); }
FB_DEV cf cadd_ib(cf a, cf b) { return cf_make(a.x - b.y, a.y + b.x); }
FB_DEV cf csub_ib(cf a, cf b) { return cf_make(a.x + b.y, a.y - b.x); }
#endif
// multiply by the direction's table twiddle: forward uses w, backward conj(w)
template <int DIR> FB_DEV cf cmul_dir(cf a, cf w) { return DIR < 0 ? cmul(a, w) : cmulc(a, w); }
// multiply by -i (forward) or +i (backward)
template <int DIR> FB_DEV cf mul_mi(cf a) { return DIR < 0 ? cf_make(a.y, -a.x) : cf_make(-a.y, a.x); }
// a + (-i b) forward, a + (i b) backward -- and the difference
template <int DIR> FB_DEV cf cadd_rot(cf a, cf b) { return DIR < 0 ? csub_ib(a, b) : cadd_ib(a, b); }
template <int DIR> FB_DEV cf csub_rot(cf a, cf b) { return DIR < 0 ? cadd_ib(a, b) : csub_ib(a, b); }

// ---- streaming ("nontemporal") global accesses --------------------------------------------------
// FB_NT is a bit mask that selects which streams carry the nt hint (tuning switch):
//   1 strided sub-pass loads   2 strided sub-pass stores   4 row-pass LDS-DMA loads   8 row-pass stores
//   16 middle kernel tendency loads   32 middle kernel derivative stores   64 middle kernel state arrays
#ifndef FB_NT
#define FB_NT 0
#endif
#ifndef FB_NT_FWD      /* nt hint on the forward strided sub-pass: 4 % on that kernel at 4096^2, nothing on the step, worse at 8192^2 */
#define FB_NT_FWD 0
#endif
typedef float f4v __attribute__((ext_vector_type(4)));
template <bool NT> FB_DEV float4 ld4(const void *p)
{
    if (NT) { const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(p)); return make_float4(v.x, v.y, v.z, v.w); }
    return *reinterpret_cast<const float4 *>(p);
}
template <bool NT> FB_DEV void st4(void *p, float4 v)
{
    if (NT) { const f4v t = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(t, reinterpret_cast<f4v *>(p)); }
    else *reinterpret_cast<float4 *>(p) = v;
}
template <bool NT> FB_DEV cf ld2(const cf *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> FB_DEV void st2(cf *p, cf v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

#define FB_SQRT1_2 0.70710678118654752440f
#define FB_C16_1 0.92387953251128675613f   /* cos(pi/8) */
#define FB_S16_1 0.38268343236508977173f   /* sin(pi/8) */

// multiply by W_16^M (forward) or its conjugate (backward), M in 0..15 compile-time
template <int M, int DIR, bool SPLIT = false> FB_DEV cf mul_w16(cf a)
{
    constexpr int m = M & 15;
    if constexpr (m == 0) return a;
    else if constexpr (m == 4) return mul_mi<DIR>(a);
    else if constexpr (m == 8) return -a;
    else if constexpr (m == 12) return mul_mi<-DIR>(a);
    else {
        // W^m = (c, -s) forward with c = cos(2 pi m/16), s = sin(2 pi m/16)
        constexpr float c = (m == 1 || m == 15) ? FB_C16_1 : (m == 2 || m == 14) ? FB_SQRT1_2 :
                            (m == 3 || m == 13) ? FB_S16_1 : (m == 5 || m == 11) ? -FB_S16_1 :
                            (m == 6 || m == 10) ? -FB_SQRT1_2 : /* 7, 9 */ -FB_C16_1;
        constexpr float s = (m == 1 || m == 7) ? FB_S16_1 : (m == 2 || m == 6) ? FB_SQRT1_2 :
                            (m == 3 || m == 5) ? FB_C16_1 : (m == 9 || m == 15) ? -FB_S16_1 :
                            (m == 10 || m == 14) ? -FB_SQRT1_2 : /* 11, 13 */ -FB_C16_1;
        constexpr float si = DIR < 0 ? -s : s;       // imaginary part of the twiddle
        return SPLIT ? cmul_k_split(a, cf_make(c, si)) : cmul_k(a, cf_make(c, si));
    }
}

// ---- in-register radix butterflies, natural-order output -----------------------------------
template <int DIR> FB_DEV void fft2(cf &a, cf &b) { cf t = a; a = cadd(t, b); b = csub(t, b); }

template <int DIR> FB_DEV void fft4(cf &a0, cf &a1, cf &a2, cf &a3)
{
    cf s0 = cadd(a0, a2), s1 = csub(a0, a2), s2 = cadd(a1, a3), d3 = csub(a1, a3);
    a0 = cadd(s0, s2); a1 = cadd_rot<DIR>(s1, d3); a2 = csub(s0, s2); a3 = csub_rot<DIR>(s1, d3);
}

template <int R, int DIR> struct Bfly;
template <int DIR> struct Bfly<1, DIR> { static FB_DEV void run(cf *) {} };
template <int DIR> struct Bfly<2, DIR> { static FB_DEV void run(cf *v) { fft2<DIR>(v[0], v[1]); } };
template <int DIR> struct Bfly<4, DIR> { static FB_DEV void run(cf *v) { fft4<DIR>(v[0], v[1], v[2], v[3]); } };
// radix 3 (grids 3*2^k, e.g. the reference's default NPTS = 768): W3 = exp(-+2 pi i/3) = (-1/2, -+sqrt(3)/2)
#define FB_SQRT3_2 0.86602540378443864676f
template <int DIR> FB_DEV void fft3(cf &a0, cf &a1, cf &a2)
{
    const cf s = cadd(a1, a2), d = csub(a1, a2);
    const cf m = cf_make(a0.x - 0.5f * s.x, a0.y - 0.5f * s.y);
    // forward: -i*(sqrt3/2)*d ; backward: +i*(sqrt3/2)*d
    const cf r = DIR < 0 ? cf_make(FB_SQRT3_2 * d.y, -FB_SQRT3_2 * d.x) : cf_make(-FB_SQRT3_2 * d.y, FB_SQRT3_2 * d.x);
    a0 = cadd(a0, s); a1 = cadd(m, r); a2 = csub(m, r);
}
template <int DIR> struct Bfly<3, DIR> { static FB_DEV void run(cf *v) { fft3<DIR>(v[0], v[1], v[2]); } };
template <int DIR> struct Bfly<8, DIR> {
    static FB_DEV void run(cf *v)
    {
        fft4<DIR>(v[0], v[2], v[4], v[6]);            // even  -> E[k] in v[0],v[2],v[4],v[6]
        __builtin_amdgcn_sched_barrier(0);
        fft4<DIR>(v[1], v[3], v[5], v[7]);            // odd   -> O[k] in v[1],v[3],v[5],v[7]
        __builtin_amdgcn_sched_barrier(0);
        cf o1 = mul_w16<2, DIR>(v[3]), o2 = v[5], o3 = mul_w16<6, DIR>(v[7]);
        cf e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6], o0 = v[1];
        v[0] = cadd(e0, o0); v[4] = csub(e0, o0);
        v[1] = cadd(e1, o1); v[5] = csub(e1, o1);
        v[2] = cadd_rot<DIR>(e2, o2); v[6] = csub_rot<DIR>(e2, o2);          // W_8^2 = -+i
        v[3] = cadd(e3, o3); v[7] = csub(e3, o3);
        __builtin_amdgcn_sched_barrier(0);
    }
};
// scheduling fence: the four independent radix-4 sub-butterflies of a radix-16 are NOT interleaved
// by the machine scheduler (one set of temporaries instead of four) -- on these register-bound
// kernels occupancy is worth more than intra-wave ILP
#define FB_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
template <int DIR> struct Bfly<16, DIR> {
    static FB_DEV void run(cf *v)
    {
        // x[4 n1 + n2]: column n2 transformed over n1
        fft4<DIR>(v[0], v[4], v[8], v[12]);  FB_SCHED_FENCE();
        fft4<DIR>(v[1], v[5], v[9], v[13]);  FB_SCHED_FENCE();
        fft4<DIR>(v[2], v[6], v[10], v[14]); FB_SCHED_FENCE();
        fft4<DIR>(v[3], v[7], v[11], v[15]); FB_SCHED_FENCE();
        // now v[4 k1 + n2] = Y_{n2}[k1]; twiddle W16^{n2 k1}
        v[5]  = mul_w16<1, DIR>(v[5]);  v[6]  = mul_w16<2, DIR>(v[6]);  v[7]  = mul_w16<3, DIR>(v[7]);
        v[9]  = mul_w16<2, DIR>(v[9]);  v[10] = mul_w16<4, DIR>(v[10]); v[11] = mul_w16<6, DIR>(v[11]);
        v[13] = mul_w16<3, DIR>(v[13]); v[14] = mul_w16<6, DIR>(v[14]); v[15] = mul_w16<9, DIR>(v[15]);
        FB_SCHED_FENCE();
        // transform over n2 for each k1 -> X[k1 + 4 k2] lands in v[4 k1 + k2]
        fft4<DIR>(v[0], v[1], v[2], v[3]);     FB_SCHED_FENCE();
        fft4<DIR>(v[4], v[5], v[6], v[7]);     FB_SCHED_FENCE();
        fft4<DIR>(v[8], v[9], v[10], v[11]);   FB_SCHED_FENCE();
        fft4<DIR>(v[12], v[13], v[14], v[15]); FB_SCHED_FENCE();
        // transpose 4x4: want v[k1 + 4 k2]
        cf t;
        t = v[1];  v[1]  = v[4];  v[4]  = t;
        t = v[2];  v[2]  = v[8];  v[8]  = t;
        t = v[3];  v[3]  = v[12]; v[12] = t;
        t = v[6];  v[6]  = v[9];  v[9]  = t;
        t = v[7];  v[7]  = v[13]; v[13] = t;
        t = v[11]; v[11] = v[14]; v[14] = t;
    }
};

// ============================================================================================
// Wave-tile FFT: one wave64 transforms 16 columns x n rows (n in 8..128) along the rows.
//
//  layout LA ("load"):  lane = (g = lane>>3, cp = lane&7); columns 2cp,2cp+1 (one float4);
//                       rows r = g + 8 m, m < n/8
//  layout LB ("freq"):  lane = (h = lane>>4, c = lane&15); column c; indices k = p + R1 q with
//                       p = h + 4 s (s < NP), q < 8, R1 = n/8; lanes with h >= R1 idle if R1 < 4
//  A2B: LA -> LB (radix-R1 in registers over m, twiddle W_n^{g p}, LDS exchange, radix-8 over g)
//  B2A: LB -> LA (radix-8 over q, twiddle, LDS exchange, radix-R1 over p)
//  LDS per wave: min(R1, 8) * WT_PSTR complex: for n = 128 (R1 = 16) the exchange runs in two rounds of 8 p-rows
//  (p = h + 4 s: rows 0-7 belong to s = 0,1 and rows 8-15 to s = 2,3), which halves the per-wave LDS (9 KB instead of 18 KB)
//  and doubles the waves a CU can hold.  Only the calling wave touches its region; DS
//  instructions of one wave execute in order, so no barrier is required between phases.
// ============================================================================================
#define WT_PSTR 144   /* [p][g(8)][col(16)] + 16 complex pad: keeps the LB accesses conflict-free */

#ifndef FB_MID128_WAVES
#define FB_MID128_WAVES 2     /* k_col_mid<128> (nx = 16384): 256 registers, two workgroups per CU */
#endif
template <int n> struct WaveTile {
    static constexpr int R1 = n / 8;
    static constexpr int NP = R1 >= 4 ? R1 / 4 : 1;
    static constexpr int NLB = NP * 8;        // complex per lane in LB
    static constexpr int NLA = n / 8;         // float4 per lane in LA
    static constexpr int ROUNDS = R1 > 8 ? R1 / 8 : 1;      // exchange rounds
    static constexpr int PR = R1 / ROUNDS;                  // p-rows per round
    static constexpr int LDS_CF = PR * WT_PSTR;
    static constexpr bool TM = R1 >= 4;       // state arrays in the tile-major layout (fb_kernels.h)
    static FB_DEV bool lb_active(int lane) { return R1 >= 4 || (lane >> 4) < R1; }
    // register-allocation target of the fused middle kernel (waves per SIMD)
    static constexpr int MID_MIN_WAVES = n >= 128 ? FB_MID128_WAVES : (n >= 64 ? 2 : (n >= 32 ? 4 : 2));   // n < 32: row-layout state path, tiny grids
};

template <int n, int DIR>
FB_DEV void wave_fft_A2B(const float4 *in /*[n/8]*/, cf *out /*[NLB]*/, cf *lds, const cf *__restrict__ tw_n, int lane)
{
    constexpr int R1 = WaveTile<n>::R1, NP = WaveTile<n>::NP;
    const int g = lane >> 3, cp = lane & 7, h = lane >> 4, c = lane & 15;
    cf a0[R1], a1[R1];
#pragma unroll
    for (int m = 0; m < R1; ++m) { a0[m] = cf_make(in[m].x, in[m].y); a1[m] = cf_make(in[m].z, in[m].w); }
    Bfly<R1, DIR>::run(a0);
    Bfly<R1, DIR>::run(a1);
    constexpr int ROUNDS = WaveTile<n>::ROUNDS, PR = WaveTile<n>::PR;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
#pragma unroll
        for (int pp = 0; pp < PR; ++pp) {
            const int p = r * PR + pp;
            if (p > 0) {
                cf w = tw_n[g * p];
                a0[p] = cmul_dir<DIR>(a0[p], w);
                a1[p] = cmul_dir<DIR>(a1[p], w);
            }
            *reinterpret_cast<float4 *>(&lds[pp * WT_PSTR + g * 16 + 2 * cp]) = make_float4(a0[p].x, a0[p].y, a1[p].x, a1[p].y);
        }
        __builtin_amdgcn_wave_barrier();
        if (WaveTile<n>::lb_active(lane)) {
#pragma unroll
            for (int s = r * (NP / ROUNDS); s < (r + 1) * (NP / ROUNDS); ++s) {
                const int pp = h + 4 * s - r * PR;              // row of p = h + 4 s within this round
                cf v[8];
#pragma unroll
                for (int gg = 0; gg < 8; ++gg) v[gg] = lds[pp * WT_PSTR + gg * 16 + c];
                Bfly<8, DIR>::run(v);
#pragma unroll
                for (int q = 0; q < 8; ++q) out[s * 8 + q] = v[q];
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <int n, int DIR>
FB_DEV void wave_fft_B2A(const cf *in /*[NLB]*/, float4 *out /*[n/8]*/, cf *lds, const cf *__restrict__ tw_n, int lane)
{
    constexpr int R1 = WaveTile<n>::R1, NP = WaveTile<n>::NP;
    const int g = lane >> 3, cp = lane & 7, h = lane >> 4, c = lane & 15;
    constexpr int ROUNDS = WaveTile<n>::ROUNDS, PR = WaveTile<n>::PR;
    cf a0[R1], a1[R1];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        if (WaveTile<n>::lb_active(lane)) {
#pragma unroll
            for (int s = r * (NP / ROUNDS); s < (r + 1) * (NP / ROUNDS); ++s) {
                const int p = h + 4 * s, pp = p - r * PR;
                cf v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = in[s * 8 + q];
                Bfly<8, DIR>::run(v);
#pragma unroll
                for (int gg = 0; gg < 8; ++gg) {
                    cf x = v[gg];
                    if (gg > 0) x = cmul_dir<DIR>(x, tw_n[p * gg]);
                    lds[pp * WT_PSTR + gg * 16 + c] = x;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int pp = 0; pp < PR; ++pp) {
            float4 t = *reinterpret_cast<const float4 *>(&lds[pp * WT_PSTR + g * 16 + 2 * cp]);
            a0[r * PR + pp] = cf_make(t.x, t.y); a1[r * PR + pp] = cf_make(t.z, t.w);
        }
        if (r + 1 < ROUNDS) __builtin_amdgcn_wave_barrier();
    }
    Bfly<R1, DIR>::run(a0);
    Bfly<R1, DIR>::run(a1);
#pragma unroll
    for (int m = 0; m < R1; ++m) out[m] = make_float4(a0[m].x, a0[m].y, a1[m].x, a1[m].y);
    __builtin_amdgcn_wave_barrier();
}

// ============================================================================================
// Workgroup FFT in LDS (row pass): Stockham autosort, 16 elements per thread, radix 4/8/16
// stages.  T = N/16 threads cooperate on one length-N FFT; element e lives at LDS slot
// pad(e) = e + (e >> 4) so that both the strided reads and the scattered writes of every
// stage are (nearly) bank-conflict free.
//
// Register distribution D_R after a last stage of radix R (and expected by a first stage of
// radix R when fed from registers):  reg[m*R + q]  <->  position (t + m*T) + q*N/R.
// ============================================================================================
FB_DEV int lds_pad(int e) { return e + (e >> 4); }
// Pacing of the strided x sub-pass: a wave's eight loads (or stores) touch rows ~1 MB apart; issued in one
// burst they measured 0.118 ms per launch at 4096^2, with 256 idle cycles between them 0.093 ms (128 and 384
// cycles: 0.106 / 0.094; 512+: slower again) -- most likely DRAM bank/channel conflicts of the large power-of-
// two-ish stride.  Applied only when the arrays are far larger than the caches (ColArgs::pace).
FB_DEV void access_gap(int pace) { if (pace) __builtin_amdgcn_s_sleep(4); }

// Opaque copy of a per-thread index: address arithmetic derived from it cannot be hoisted out
// of the enclosing loop (LICM would otherwise keep dozens of invariant addresses live in VGPRs).
FB_DEV int launder(int v) { asm volatile("" : "+v"(v)); return v; }

// Workgroup barrier that orders LDS traffic only: waits lgkmcnt(0), not vmcnt, so LDS-DMA
// prefetches and global stores stay in flight across it (cdna_hip_programming.md, section 5).
#ifdef FB_NO_LDS_BARRIER   /* timing experiment only (results are wrong): what the workgroup barriers cost */
FB_DEV void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
#else
FB_DEV void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#endif

// Row plans: radices of the BACKWARD (c2r) row transform; the forward transform walks them
// reversed, so that the backward pass's final register distribution feeds the forward pass.
// Plans up to 4096 are palindromes: one stage-twiddle table (kept in registers) serves both
// directions.  Larger rows stream their twiddles from the (L2-resident) table instead.
template <int N> struct RowPlan;
template <> struct RowPlan<64>    { static constexpr int S = 2; static constexpr int R[4] = {8, 8, 1, 1}; };
template <> struct RowPlan<128>   { static constexpr int S = 3; static constexpr int R[4] = {4, 8, 4, 1}; };
template <> struct RowPlan<256>   { static constexpr int S = 2; static constexpr int R[4] = {16, 16, 1, 1}; };
template <> struct RowPlan<512>   { static constexpr int S = 3; static constexpr int R[4] = {8, 8, 8, 1}; };
template <> struct RowPlan<1024>  { static constexpr int S = 3; static constexpr int R[4] = {8, 16, 8, 1}; };
template <> struct RowPlan<2048>  { static constexpr int S = 3; static constexpr int R[4] = {16, 8, 16, 1}; };
template <> struct RowPlan<4096>  { static constexpr int S = 3; static constexpr int R[4] = {16, 16, 16, 1}; };
template <> struct RowPlan<8192>  { static constexpr int S = 4; static constexpr int R[4] = {4, 8, 16, 16}; };
template <> struct RowPlan<16384> { static constexpr int S = 4; static constexpr int R[4] = {8, 16, 16, 8}; };
template <int N> struct RowRes { static constexpr bool value = true; };   // last-stage twiddles in registers, middle stages in an LDS table

// register order of a radix-R stage: reg[e], e = m*R + q  <->  position t + ord_i<R>(e) * (N/16)
template <int R> constexpr int ord_i(int e) { return (e / R) + (16 / R) * (e % R); }

// Twiddle table offsets (in complex elements) per stage for a plan walked in the given order.
// Stage s (s >= 1) stores (16/R)*(R-1)*T entries; stage 0 has NS = 1 and stores nothing.
template <int N, bool FWD> struct RowTw {
    static constexpr int T = N / 16;
    static constexpr int radix(int s) { return FWD ? RowPlan<N>::R[RowPlan<N>::S - 1 - s] : RowPlan<N>::R[s]; }
    static constexpr int ns(int s) { int v = 1; for (int i = 0; i < s; ++i) v *= radix(i); return v; }
    static constexpr int size(int s) { return s == 0 ? 0 : (16 / radix(s)) * (radix(s) - 1) * T; }
    static constexpr int offset(int s) { int v = 0; for (int i = 0; i < s; ++i) v += size(i); return v; }
    static constexpr int total() { return offset(RowPlan<N>::S); }
};

// true when the forward plan (reversed radices) equals the backward plan
template <int N> struct RowPlanSymmetric {
    static constexpr bool value = []() {
        for (int s = 0; s < RowPlan<N>::S; ++s) if (RowTw<N, true>::radix(s) != RowTw<N, false>::radix(s)) return false;
        return true; }();
};

// Twiddle source of one direction.  Table layout (host, make_row_table): offset(s) + e*T + t for
// entry e = m*(R-1) + (q-1) of stage s; the value depends on (j % NS, q) only, j = t + m*T.
//   RES : the LAST stage's values (unique per thread) live in registers, loaded once per
//         workgroup; the middle stages' few distinct values (NS*(R-1) <= 240) come from a small
//         LDS table shared by the workgroup.  Keeps the FFT at ~160 VGPRs.
//   !RES: everything streamed from the (L2-resident) global table.
template <int N, bool FWD, bool RES> struct RowTwSrc;
template <int N, bool FWD> struct RowTwSrc<N, FWD, true> {
    using TW = RowTw<N, FWD>;
    static constexpr int S = RowPlan<N>::S, T = N / 16;
    static constexpr int per(int s) { return (16 / TW::radix(s)) * (TW::radix(s) - 1); }
    // LDS table: stages 1..S-2, stage s holds ns(s)*(R_s-1) entries [k][q-1]
    static constexpr int lsize(int s) { return (s >= 1 && s <= S - 2) ? TW::ns(s) * (TW::radix(s) - 1) : 0; }
    static constexpr int loff(int s) { int v = 0; for (int i = 1; i < s; ++i) v += lsize(i); return v; }
    static constexpr int LDS_CF = loff(S - 1) > 0 ? ((loff(S - 1) + 1) / 2) * 2 : 0;
    cf w[per(S - 1)];
    const cf *ltab; int t;
    // cooperative: fills the LDS table (all threads of the workgroup), then the registers
    FB_DEV void init(const cf *__restrict__ tab, cf *lds_tab, int t_, int tid, int nthreads)
    {
        t = t_; ltab = lds_tab;
        if constexpr (S >= 3) fill<1>(tab, lds_tab, tid, nthreads);
        if constexpr (S >= 4) fill<2>(tab, lds_tab, tid, nthreads);
#pragma unroll
        for (int e = 0; e < per(S - 1); ++e) w[e] = tab[TW::offset(S - 1) + e * T + t_];
    }
    template <int s> FB_DEV void fill(const cf *__restrict__ tab, cf *lds_tab, int tid, int nthreads)
    {
        constexpr int R = TW::radix(s), NS = TW::ns(s);
        for (int i = tid; i < NS * (R - 1); i += nthreads) {
            const int k = i / (R - 1), qm = i - k * (R - 1);      // j = k (m = 0, t = k < NS <= T)
            lds_tab[loff(s) + i] = tab[TW::offset(s) + qm * T + k];
        }
    }
    template <int s> FB_DEV cf get(int e) const
    {
        if constexpr (s == S - 1) return w[e];
        else {
            constexpr int R = TW::radix(s), NS = TW::ns(s);
            const int m = e / (R - 1), qm = e - m * (R - 1);
            return ltab[loff(s) + ((t + m * T) % NS) * (R - 1) + qm];
        }
    }
};
template <int N, bool FWD> struct RowTwSrc<N, FWD, false> {
    using TW = RowTw<N, FWD>;
    static constexpr int LDS_CF = 0;
    const cf *__restrict__ tab; int t;
    FB_DEV void init(const cf *__restrict__ tab_, cf *, int t_, int, int) { tab = tab_; t = t_; }
    template <int s> FB_DEV cf get(int e) const { return tab[TW::offset(s) + e * TW::T + t]; }
};

// One Stockham stage (stage index SI of its plan).  LDS addresses are written as
// (per-thread base) + (compile-time offset) wherever the padding is linear in q, so that the DS
// instructions carry immediate offsets instead of one address register per element.
template <int N, int R, int NS, int DIR, bool FROM_REGS, bool TO_REGS, int SI, class SRC>
FB_DEV void stockham_stage(cf *lds, int t, const SRC &src, cf *reg /*[16]*/)
{
    constexpr int T = N / 16, NB = 16 / R, STR = N / R;
    constexpr bool RD_LIN = (STR % 16) == 0;                           // pad(j + q STR) = pad(j) + q (STR + STR/16)
    constexpr bool WR_LIN = ((NS * R) % 16 == 0) && (NS % 16 == 0 || NS * R <= 16);
#pragma unroll
    for (int m = 0; m < NB; ++m) {
        const int j = t + m * T;
        cf *v = reg + m * R;
        if (!FROM_REGS) {
            const cf *rb = lds + lds_pad(j);
#pragma unroll
            for (int q = 0; q < R; ++q) v[q] = lds_rd(RD_LIN ? &rb[q * (STR + STR / 16)] : &lds[lds_pad(j + q * STR)]);
        }
        if (NS > 1) {
#pragma unroll
            for (int q = 1; q < R; ++q) v[q] = cmul_dir<DIR>(v[q], src.template get<SI>(m * (R - 1) + (q - 1)));
        }
        Bfly<R, DIR>::run(v);
    }
    if (!TO_REGS) {
        lds_barrier();                            // previous readers of this buffer are done
#pragma unroll
        for (int m = 0; m < NB; ++m) {
            const int j = t + m * T;
            const int j0 = (j / NS) * NS * R + (j % NS);
            cf *wb = lds + lds_pad(j0);
#pragma unroll
            for (int q = 0; q < R; ++q) {
                if (WR_LIN) lds_wr(&wb[q * NS + (q * NS) / 16], reg[m * R + q]);
                else lds_wr(&lds[lds_pad(j0 + q * NS)], reg[m * R + q]);
            }
        }
        lds_barrier();
    }
}

// Whole transform, registers -> registers.  Input order: first stage's (ord_i<R_first>), output
// order: last stage's (ord_i<R_last>).  FWD selects direction and plan order.
template <int N, bool FWD, class SRC>
FB_DEV void row_fft(cf *lds, int t, const SRC &src, cf *reg)
{
    using TW = RowTw<N, FWD>;
    constexpr int S = RowPlan<N>::S, DIR = FWD ? -1 : +1;
    stockham_stage<N, TW::radix(0), 1, DIR, true, false, 0>(lds, launder(t), src, reg);
    if constexpr (S == 2) {
        stockham_stage<N, TW::radix(1), TW::ns(1), DIR, false, true, 1>(lds, launder(t), src, reg);
    } else if constexpr (S == 3) {
        stockham_stage<N, TW::radix(1), TW::ns(1), DIR, false, false, 1>(lds, launder(t), src, reg);
        stockham_stage<N, TW::radix(2), TW::ns(2), DIR, false, true, 2>(lds, launder(t), src, reg);
    } else {
        stockham_stage<N, TW::radix(1), TW::ns(1), DIR, false, false, 1>(lds, launder(t), src, reg);
        stockham_stage<N, TW::radix(2), TW::ns(2), DIR, false, false, 2>(lds, launder(t), src, reg);
        stockham_stage<N, TW::radix(3), TW::ns(3), DIR, false, true, 3>(lds, launder(t), src, reg);
    }
}


// ============================================================================================
// Half-buffer variant of the row transform: the LDS exchange between two stages is done in two
// rounds over a buffer of N/2 elements (positions [0,N/2) first, then [N/2,N)), which halves the
// exchange LDS of a workgroup (more workgroups per CU) for two extra barriers per exchange.
// Writers of stage s land in the first half iff their butterfly index j < N/(2R); readers of
// stage s+1 take their inputs q' < R'/2 from the first half.
// ============================================================================================
template <int N, int R, int NS, int DIR, int SI, class SRC>
FB_DEV void stage_compute(int t, const SRC &src, cf *reg)
{
    constexpr int T = N / 16, NB = 16 / R;
#pragma unroll
    for (int m = 0; m < NB; ++m) {
        cf *v = reg + m * R;
        if (NS > 1) {
#pragma unroll
            for (int q = 1; q < R; ++q) v[q] = cmul_dir<DIR>(v[q], src.template get<SI>(m * (R - 1) + (q - 1)));
        }
        Bfly<R, DIR>::run(v);
    }
    (void)T; (void)t;
}

// exchange: outputs of a radix-R stage with stride NS  ->  inputs of the next stage of radix R2
template <int N, int R, int NS, int R2>
FB_DEV void stage_exchange_half(cf *lds, int t, cf *reg)
{
    constexpr int T = N / 16, NB = 16 / R, NB2 = 16 / R2, STR2 = N / R2, HALF = N / 2;
    cf out[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) out[e] = reg[e];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        lds_barrier();                                              // previous readers are done
#pragma unroll
        for (int m = 0; m < NB; ++m) {
            const int j = t + m * T;
            if ((j < N / (2 * R)) == (half == 0)) {
                const int j0 = (j / NS) * NS * R + (j % NS) - half * HALF;
#pragma unroll
                for (int q = 0; q < R; ++q) lds[lds_pad(j0 + q * NS)] = out[m * R + q];
            }
        }
        lds_barrier();
#pragma unroll
        for (int m = 0; m < NB2; ++m) {
            const int j = t + m * T;
#pragma unroll
            for (int q = 0; q < R2 / 2; ++q) {
                const int qq = q + half * (R2 / 2);
                reg[m * R2 + qq] = lds[lds_pad(j + qq * STR2 - half * HALF)];
            }
        }
    }
}

template <int N, bool FWD, class SRC>
FB_DEV void row_fft_half(cf *lds, int t, const SRC &src, cf *reg)
{
    using TW = RowTw<N, FWD>;
    constexpr int S = RowPlan<N>::S, DIR = FWD ? -1 : +1;
    static_assert(S == 3, "half-buffer row transform is instantiated for three-stage plans");
    stage_compute<N, TW::radix(0), 1, DIR, 0>(launder(t), src, reg);
    stage_exchange_half<N, TW::radix(0), 1, TW::radix(1)>(lds, launder(t), reg);
    stage_compute<N, TW::radix(1), TW::ns(1), DIR, 1>(launder(t), src, reg);
    stage_exchange_half<N, TW::radix(1), TW::ns(1), TW::radix(2)>(lds, launder(t), reg);
    stage_compute<N, TW::radix(2), TW::ns(2), DIR, 2>(launder(t), src, reg);
}
