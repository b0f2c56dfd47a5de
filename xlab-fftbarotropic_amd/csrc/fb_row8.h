// fb_row8.h -- fused row pass for ny = 4096 = 8^4 with 512 threads per row pair, 8 elements per thread.
//
// Why a second row kernel: the Stockham kernel (k_row) keeps 16 elements per thread, needs ~200 VGPRs and
// so runs 2 waves per SIMD with 6 workgroup barriers per transform; measured without any HBM traffic it
// still takes 75 us per launch at 4096^2 (the memory floor is 59 us) -- a latency chain, not a throughput
// limit.  Here the 4096-point transform is a decimation-in-frequency (backward) / decimation-in-time
// (forward) pair over the index digits  position = d0 + 8 d1 + 64 d2 + 512 d3:
//     thread = (wave w, lane = 8 l_hi + l_lo),  8 registers e
//     stage 0: digit d3 in registers (positions t + 512 e)        twiddle W_4096^{p t}
//     G exchange  (registers <-> wave index)      through LDS, the only workgroup-wide one (2 barriers)
//     stage 1: digit d2                                            twiddle W_512^{p lane}
//     A exchange  (registers <-> l_hi)            wave-private LDS slice, no barrier
//     stage 2: digit d1                                            twiddle W_64^{p l_lo}
//     B exchange  (registers <-> l_lo)            wave-private
//     stage 3: digit d0
// The backward output (physical space) comes out digit-reversed, y = w + 8 l_hi + 64 l_lo + 512 e -- the
// Jacobian is pointwise, so the order does not matter -- and the forward transform is the transposed
// sequence, which takes that order in and leaves the spectrum in natural order for the untangle/store.
// ~100 VGPRs -> 4 waves per SIMD; 4 workgroup barriers per transform instead of 6 (8 waves each).
// Same arithmetic as k_row<4096, ROW_FUSED> up to rounding order (main.cpp:154-237, y part).
#pragma once
#include "fb_kernels.h"

struct Row8 {
    static constexpr int N = 4096, T = 512, E = 8;
    static constexpr int SLICE = 576;                 // complex per wave slice: 8 rows of pitch <= 72 (A/B exchanges), 512 used by G
    // LDS has 32 four-byte banks: a 64-bit access is conflict-free when each 16 consecutive lanes hit 16
    // different complex slots mod 16.  Writers store row e at e*PITCH + lane; the l_hi swap reads
    // l_hi*72 + 8 e' + l_lo (pitch = 8 mod 16), the l_lo swap reads l_lo*65 + 8 l_hi + e' (pitch = 1 mod 16).
    static constexpr int PITCH_HI = 72, PITCH_LO = 65;
    static constexpr int XBUF = 8 * SLICE, STG = N + 2; // exchange buffer, staging [A: N/2][B: N/2][A(N/2)][B(N/2)]
    static constexpr int TW2 = 64;                    // W_64^{p l_lo} at [p][l_lo]
    static constexpr size_t LDS_BYTES = (size_t)(XBUF + STG + TW2) * sizeof(cf);
};

// swap the register index with the wave index (workgroup-wide)
template <bool LEAD = true> FB_DEV void r8_xch_group(cf *v, cf *xbuf, int w, int l)
{
#ifdef FB_R8_NOXG   /* timing experiment only */
    return;
#endif
    if (LEAD) lds_barrier();                          // every wave is done with its slice (LEAD = false: the caller has had a workgroup barrier since the slices were last read -- the backward transforms, which start right behind the staging barriers of their phase)
#pragma unroll
    for (int p = 0; p < 8; ++p) lds_wr(&xbuf[p * Row8::SLICE + w * 64 + l], v[p]);
    lds_barrier();
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = lds_rd(&xbuf[w * Row8::SLICE + e * 64 + l]);
}
// swap the register index with l_hi (HI) or l_lo (!HI) inside the wave's own slice
template <bool HI> FB_DEV void r8_xch_wave(cf *v, cf *slice, int l_hi, int l_lo)
{
#ifdef FB_R8_NOXW   /* timing experiment only */
    return;
#endif
    constexpr int PITCH = HI ? Row8::PITCH_HI : Row8::PITCH_LO;
    cf *wr = slice + l_hi * 8 + l_lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) lds_wr(&wr[e * PITCH], v[e]);
    __builtin_amdgcn_wave_barrier();
    const cf *rd = HI ? slice + l_hi * PITCH + l_lo : slice + l_lo * PITCH + l_hi * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = lds_rd(&rd[e * (HI ? 8 : 1)]);
    __builtin_amdgcn_wave_barrier();
}

// W_4096^{p t} and W_512^{p lane} in registers, W_64^{p l_lo} in a 512-byte LDS table; p = 1..7, forward sign
struct Row8Tw { cf w0[7], w1[7]; const cf *w2; };

// backward (DIR = +1): natural order in (position t + 512 e), digit-reversed out
FB_DEV void r8_bwd(cf *v, cf *xbuf, const Row8Tw &tw, int w, int l)
{
    const int l_hi = l >> 3, l_lo = l & 7;
    cf *slice = xbuf + w * Row8::SLICE;
    Bfly<8, +1>::run(v);
#pragma unroll
    for (int p = 1; p < 8; ++p) v[p] = cmulc(v[p], tw.w0[p - 1]);
    r8_xch_group<false>(v, xbuf, w, l);
    Bfly<8, +1>::run(v);
#pragma unroll
    for (int p = 1; p < 8; ++p) v[p] = cmulc(v[p], tw.w1[p - 1]);
    r8_xch_wave<true>(v, slice, l_hi, l_lo);
    Bfly<8, +1>::run(v);
#pragma unroll
    for (int p = 1; p < 8; ++p) v[p] = cmulc(v[p], lds_rd(&tw.w2[p * 8 + l_lo]));
    r8_xch_wave<false>(v, slice, l_hi, l_lo);
    Bfly<8, +1>::run(v);
}
// forward (DIR = -1): the transposed sequence, digit-reversed in, natural order out
FB_DEV void r8_fwd(cf *v, cf *xbuf, const Row8Tw &tw, int w, int l)
{
    const int l_hi = l >> 3, l_lo = l & 7;
    cf *slice = xbuf + w * Row8::SLICE;
    Bfly<8, -1>::run(v);
    r8_xch_wave<false>(v, slice, l_hi, l_lo);
#pragma unroll
    for (int p = 1; p < 8; ++p) v[p] = cmul(v[p], lds_rd(&tw.w2[p * 8 + l_lo]));
    Bfly<8, -1>::run(v);
    r8_xch_wave<true>(v, slice, l_hi, l_lo);
#pragma unroll
    for (int p = 1; p < 8; ++p) v[p] = cmul(v[p], tw.w1[p - 1]);
    Bfly<8, -1>::run(v);
    r8_xch_group(v, xbuf, w, l);
#pragma unroll
    for (int p = 1; p < 8; ++p) v[p] = cmul(v[p], tw.w0[p - 1]);
    Bfly<8, -1>::run(v);
}

// LDS-DMA prefetch of two half-spectrum rows: stg[0..N/2) = row A, stg[N/2..N) = row B, then the two Nyquist
// elements (also by DMA, one dword per lane: a register load here would be waited for on the spot by the one
// wave that issues it, and the whole workgroup waits for that wave at the next barrier)
template <bool SLAB>
FB_DEV void r8_dma_issue(cf *stg, int t, const RowView &v, int fA, int fB, int row)
{
    constexpr int N = Row8::N;
    const int w = t >> 6, lane = t & 63;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int ch = w + c * 8, k = ch * 128 + lane * 2;        // 16 chunks of 1 KiB per row, two per wave
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)row_ptr<SLAB>(v, fA, row, k),
                                         (void __attribute__((address_space(3))) *)(stg + ch * 128), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)row_ptr<SLAB>(v, fB, row, k),
                                         (void __attribute__((address_space(3))) *)(stg + N / 2 + ch * 128), 16, 0, 0);
    }
    if (t < 4) {                                                  // lanes 0,1: A(N/2).x,.y  lanes 2,3: B(N/2).x,.y
        const float *src = reinterpret_cast<const float *>(row_ptr<SLAB>(v, t < 2 ? fA : fB, row, N / 2)) + (t & 1);
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)src,
                                         (void __attribute__((address_space(3))) *)(stg + N), 4, 0, 0);
    }
}

// Wait for the rows of the phase about to start.  (Tried and dropped: touching the rows of the phase AFTER
// next, one dword per 64 bytes with the wait relaxed to vmcnt(1), so that the LDS-DMA finds them in L2 --
// 0.089 ms against 0.083 ms per launch at 4096^2: the extra requests cost more than the longer lead gains.)
#define R8_WAIT_ROWS() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

// Hermitian extension Z = A_ext + i B_ext into the first backward stage's registers (SURVEY note N2)
FB_DEV void r8_ext(cf *v, int t, const cf *stg)
{
#ifdef FB_R8_NOEXT  /* timing experiment only */
    for (int e = 0; e < 8; ++e) v[e] = cf_make(1.f + e, 1.f + t);
    return;
#endif
    constexpr int N = Row8::N, T = Row8::T;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        if (e < 4) {
            const int k = t + e * T;
            const cf a = lds_rd(&stg[k]), b = lds_rd(&stg[N / 2 + k]);
            v[e] = (e == 0 && t == 0) ? cf_make(a.x, b.x) : cf_make(a.x - b.y, a.y + b.x);
        } else {
            const int k = (8 - e) * T - t;                         // mirror, in (0, N/2]
            const bool nyq = (e == 4 && t == 0);                   // k = N/2: the Nyquist slots behind the two rows
            const cf a = lds_rd(&stg[nyq ? N : k]), b = lds_rd(&stg[nyq ? N + 1 : N / 2 + k]);
            v[e] = nyq ? cf_make(a.x, b.x) : cf_make(a.x + b.y, b.x - a.y);
        }
    }
}

template <bool SLAB>
__global__ void __launch_bounds__(512, 4) k_row8(RowArgs a, const cf *__restrict__ root /* W_4096^j */)
{
    constexpr int N = Row8::N, T = Row8::T;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf *xbuf = reinterpret_cast<cf *>(smem_raw);
    cf *stg = xbuf + Row8::XBUF;
    const int t = threadIdx.x, w = t >> 6, l = t & 63;
    Row8Tw tw;
#pragma unroll
    for (int p = 1; p < 8; ++p) { tw.w0[p - 1] = root[p * t]; tw.w1[p - 1] = root[8 * p * l]; }
    cf *tw2 = stg + Row8::STG;
    if (t < 64) tw2[t] = root[64 * (t & 7) * (t >> 3)];          // [p = t >> 3][l_lo = t & 7]
    tw.w2 = tw2;
    // land the twiddle loads here: a wait the compiler places at their first use inside the loop would be a
    // vmcnt(0) right behind the prefetch of the next rows
#pragma unroll
    for (int p = 0; p < 7; ++p) asm volatile("" :: "v"(tw.w0[p]), "v"(tw.w1[p]));
    __syncthreads();

    const int npairs = a.nx >> 1;
    const int iters = (npairs + gridDim.x - 1) / gridDim.x;
#ifdef FB_ROW_SAMEROW   /* timing experiment only: every workgroup works on rows 0,1 (no HBM traffic); results are wrong */
    auto pair_of = [&](int it, bool &valid) { const int pr = it * gridDim.x + blockIdx.x; valid = pr < npairs; return a.x0; };
#else
    auto pair_of = [&](int it, bool &valid) { const int pr = it * gridDim.x + blockIdx.x; valid = pr < npairs; return a.x0 + (valid ? 2 * pr : 0); };
#endif
    if (iters > 0) {
        bool v; const int x = pair_of(0, v);
        r8_dma_issue<SLAB>(stg, t, a.M, 0, 1, x);
        R8_WAIT_ROWS();
    }
    for (int it = 0; it < iters; ++it) {
        bool valid;
        const int x0 = pair_of(it, valid), x1 = x0 + 1;           // an invalid workgroup recomputes pair 0, stores nothing
        cf v[8];
        float t0[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) t0[e] = 0.f;
#pragma unroll 1
        for (int r = 0; r < 2; ++r) {
            const int x = x0 + r;
            const int tt = launder(t), wl = tt >> 6, ll = tt & 63;
            float zx[8], zy[8];
            // ---- dvortdx, dvortdy of row x (vmcnt also counts stores: row (0,0) was waited for before the previous stores)
            if (r == 1) R8_WAIT_ROWS();
            lds_barrier();
            r8_ext(v, tt, stg);
            lds_barrier();
            r8_dma_issue<SLAB>(stg, tt, a.M, 2, 3, x);
            r8_bwd(v, xbuf, tw, wl, ll);
#pragma unroll
            for (int e = 0; e < 8; ++e) { zx[e] = v[e].x * a.scale; zy[e] = v[e].y * a.scale; }   // main.cpp:154,168
            // ---- u, v of row x
            R8_WAIT_ROWS();
            lds_barrier();
            r8_ext(v, tt, stg);
            lds_barrier();
            bool vn = true; int xn = x1;
            if (r == 1) xn = (it + 1 < iters) ? pair_of(it + 1, vn) : -1;
            if (xn >= 0) r8_dma_issue<SLAB>(stg, tt, a.M, 0, 1, xn);
            r8_bwd(v, xbuf, tw, wl, ll);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float u = -(v[e].x * a.scale);              // main.cpp:200-201
                const float vv = v[e].y * a.scale;                // main.cpp:214
                v[e].y = -u * zx[e] - vv * zy[e];                 // main.cpp:225-227 ...
            }
            if (a.src) {                                          // ... + vort_src; its loads (and their wait) stay in this branch
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e].y += a.src[(size_t)x * N + wl + 8 * (ll >> 3) + 64 * (ll & 7) + 512 * e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float val = v[e].y;
                v[e] = cf_make(t0[e], val);                       // complete only after r == 1
                t0[e] = val;
            }
        }
        const int tt = launder(t), wl = tt >> 6, ll = tt & 63;
        r8_fwd(v, xbuf, tw, wl, ll);                              // main.cpp:237 (y part), rows x0 and x1 packed
        R8_WAIT_ROWS();                                           // the next pair's first rows have landed
        // untangle Z = FFT(t0 + i t1): the upper half (e >= 4) goes through LDS, position k - N/2
        lds_barrier();
#pragma unroll
        for (int e = 4; e < 8; ++e) xbuf[tt + (e - 4) * T] = v[e];
        lds_barrier();
#ifdef FB_R8_NOST   /* timing experiment only: (almost) no stores */
        if (valid && v[0].x == 123.456f) {
#else
        if (valid) {
#endif
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = tt + e * T;
                const cf zk = v[e];
                const cf zn = (e == 0 && tt == 0) ? zk : lds_rd(&xbuf[N / 2 - k]);            // Z[N - k] sits at N - k - N/2
                if (!row_keep<SLAB>(a.T, a.t_frozen, k)) continue;
                st2<(FB_NT & 8) != 0>(const_cast<cf *>(row_ptr<SLAB>(a.T, 0, x0, k)), cf_make(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y)));
                st2<(FB_NT & 8) != 0>(const_cast<cf *>(row_ptr<SLAB>(a.T, 0, x1, k)), cf_make(0.5f * (zk.y + zn.y), 0.5f * (zn.x - zk.x)));
            }
            if (tt == 0 && row_keep<SLAB>(a.T, a.t_frozen, N / 2)) {  // Nyquist: its own mirror
                *const_cast<cf *>(row_ptr<SLAB>(a.T, 0, x0, N / 2)) = cf_make(v[4].x, 0.f);
                *const_cast<cf *>(row_ptr<SLAB>(a.T, 0, x1, N / 2)) = cf_make(v[4].y, 0.f);
            }
        }
    }
}
