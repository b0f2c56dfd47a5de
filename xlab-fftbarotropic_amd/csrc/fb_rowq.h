// fb_rowq.h -- fused row pass for ny = 4096 with ONE real row per 256-thread workgroup: four contexts per CU.
//
// k_row8 (fb_row8.h) packs two real rows into one 4096-point complex transform run by 512 threads: two workgroups per CU, every
// phase fenced by eight-wave barriers.  Here one real row of N = 4096 points is one complex transform of M = 2048 points
// (even/odd packing, exactly as in fb_rowh.h), run by FOUR waves: 35 KB of LDS per workgroup, four workgroups per CU, so that
// four rows in different phases share a CU and a barrier only ever stops four waves.  M = 2048 = 8 x 4 x 8 x 8:
//     thread = (wave w 0..3, lane = 8 l_hi + l_lo), 8 registers
//     stage 0: radix 8 over the registers (positions t + 256 e)              twiddle W_2048^{p t}
//     G exchange (registers <-> wave index), the only workgroup-wide one: afterwards wave w' holds p = w' and p = w' + 4
//     stage 1: two radix-4 butterflies over the old wave index              twiddle W_256^{q lane}
//     A exchange (registers <-> l_hi, wave-private), stage 2: radix 8       twiddle W_64^{r l_lo}
//     B exchange (registers <-> l_lo, wave-private), stage 3: radix 8
// backward (decimation in frequency) as listed, forward the transposed sequence.  The backward output is digit-reversed:
//     j = (w + 4 (l_hi >> 2)) + 8 (l_hi & 3) + 32 l_lo + 256 e
// which never matters (the Jacobian is pointwise; vort_src is permuted once into this order).
// Same arithmetic as k_row8 / k_row<4096, ROW_FUSED> up to rounding order (main.cpp:154-237, y part).
#pragma once
#include "fb_rowh.h"

#ifndef RQ_NT      /* nontemporal hint: 1 = the LDS-DMA loads of the four fields (read once: streamed, they leave the tendency rows in the caches for
                      k_col_full -- row pass -1.6 %, k_col_full -1.7 %), 2 = the tendency stores (k_col_full +1 %) */
#define RQ_NT 1
#endif
// Opaque thread id per phase (launder): fewer live registers, but every phase recomputes its LDS addresses.  The one-GPU instances have
// the registers to do without (118-124 of 128): 217 of 381 non-packed vector instructions per row gone, -2.5 % per launch.  The slab-blocked
// instances (more address arithmetic) would spill and keep it.  RQ_LAUNDER_MODE: 3 = always, 0 = never, unset = by instance.
#ifdef RQ_LAUNDER_MODE
#define RQ_LAUNDER(t) ((RQ_LAUNDER_MODE & 2) ? launder(t) : (t))
#define RQ_LAUNDER2(t) ((RQ_LAUNDER_MODE & 1) ? launder(t) : (t))
#else
#define RQ_LAUNDER(t) (SLAB ? launder(t) : (t))
#define RQ_LAUNDER2(t) (SLAB ? launder(t) : (t))
#endif
struct RowQ {
    static constexpr int M = 2048, N = 4096, T = 256;
    static constexpr int SLICE = Row8::SLICE;                  // per-wave slice of the exchange buffer (A/B exchanges)
    static constexpr int XBUF = 4 * SLICE;                      // four wave slices; the G exchange uses 8 * 64 of each
    static constexpr int STG = M + 2;
    static constexpr int TW2 = 64;
    static constexpr size_t LDS_BYTES = (size_t)(XBUF + STG + TW2) * sizeof(cf);
};
struct RowQTw { cf w0[7], w1[3]; const cf *w2; };              // W_2048^{p t}, W_256^{q lane} in registers; W_64^{r l_lo} in LDS

// G exchange.  Element p of the stage-0 output of wave w goes to wave p & 3 as register (q = w, h = p >> 2); it travels through
// the RECEIVING wave's slice, [p & 3][(p >> 2) * 4 + w][lane], so that after the barrier every wave reads its own slice only and
// may go on to the wave-private exchanges without another barrier.
template <class F> FB_DEV void rq_xch_group_bwd(cf *v, cf *xbuf, int w, int l, F &&behind_barrier)
{
#ifdef FB_R8_NOXG   /* timing experiment only */
    behind_barrier(); return;
#endif
    // no barrier in front: the slices were last read by the previous transform's wave-private exchanges, and the caller has passed the
    // staging barriers of this phase since
#pragma unroll
    for (int p = 0; p < 8; ++p) lds_wr(&xbuf[(p & 3) * RowQ::SLICE + ((p >> 2) * 4 + w) * 64 + l], v[p]);
    lds_barrier();
    behind_barrier();                                 // every wave has also left the staged row behind: the next one may be sent for
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = lds_rd(&xbuf[w * RowQ::SLICE + j * 64 + l]);         // j = 4 h + q
}
// forward direction: the inverse permutation (register (q, h) of wave w is element p = w + 4 h of wave q)
FB_DEV void rq_xch_group_fwd(cf *v, cf *xbuf, int w, int l)
{
#ifdef FB_R8_NOXG   /* timing experiment only */
    return;
#endif
#pragma unroll
    for (int j = 0; j < 8; ++j) lds_wr(&xbuf[w * RowQ::SLICE + j * 64 + l], v[j]);          // own slice: no barrier needed before
    lds_barrier();
#pragma unroll
    for (int p = 0; p < 8; ++p) v[p] = lds_rd(&xbuf[(p & 3) * RowQ::SLICE + ((p >> 2) * 4 + w) * 64 + l]);
    lds_barrier();                                    // the slices are free again
}

template <class F> FB_DEV void rq_bwd(cf *v, cf *xbuf, const RowQTw &tw, int w, int l, F &&behind_barrier)
{
    const int l_hi = l >> 3, l_lo = l & 7;
    cf *slice = xbuf + w * RowQ::SLICE;
    Bfly<8, +1>::run(v);
#pragma unroll
    for (int p = 1; p < 8; ++p) v[p] = cmulc(v[p], tw.w0[p - 1]);
    rq_xch_group_bwd(v, xbuf, w, l, behind_barrier);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        fft4<+1>(v[4 * h], v[4 * h + 1], v[4 * h + 2], v[4 * h + 3]);
#pragma unroll
        for (int q = 1; q < 4; ++q) v[4 * h + q] = cmulc(v[4 * h + q], tw.w1[q - 1]);
    }
    r8_xch_wave<true>(v, slice, l_hi, l_lo);
    Bfly<8, +1>::run(v);
#pragma unroll
    for (int p = 1; p < 8; ++p) v[p] = cmulc(v[p], lds_rd(&tw.w2[p * 8 + l_lo]));
    r8_xch_wave<false>(v, slice, l_hi, l_lo);
    Bfly<8, +1>::run(v);
}

FB_DEV void rq_fwd(cf *v, cf *xbuf, const RowQTw &tw, int w, int l)
{
    const int l_hi = l >> 3, l_lo = l & 7;
    cf *slice = xbuf + w * RowQ::SLICE;
    Bfly<8, -1>::run(v);
    r8_xch_wave<false>(v, slice, l_hi, l_lo);
#pragma unroll
    for (int p = 1; p < 8; ++p) v[p] = cmul(v[p], lds_rd(&tw.w2[p * 8 + l_lo]));
    Bfly<8, -1>::run(v);
    r8_xch_wave<true>(v, slice, l_hi, l_lo);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int q = 1; q < 4; ++q) v[4 * h + q] = cmul(v[4 * h + q], tw.w1[q - 1]);
        fft4<-1>(v[4 * h], v[4 * h + 1], v[4 * h + 2], v[4 * h + 3]);
    }
    rq_xch_group_fwd(v, xbuf, w, l);
#pragma unroll
    for (int p = 1; p < 8; ++p) v[p] = cmul(v[p], tw.w0[p - 1]);
    Bfly<8, -1>::run(v);
}

template <bool SLAB>
FB_DEV void rq_dma_issue(cf *stg, int t, const RowView &view, int field, int row)
{
    constexpr int M = RowQ::M;
    const int w = t >> 6, lane = t & 63;
#pragma unroll
    for (int c = 0; c < 4; ++c) {                     // 16 chunks of 1 KiB, 4 per wave
        const int ch = w + c * 4, k = ch * 128 + lane * 2;
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)row_ptr<SLAB>(view, field, row, k), rh_to_lds(stg + ch * 128), 16, 0, (RQ_NT & 1) ? 2 : 0);
    }
    rh_lds_ptr nyq = rh_to_lds(stg + M);              // X[M]: one dword per lane (lanes 0, 1)
    const float *src = reinterpret_cast<const float *>(row_ptr<SLAB>(view, field, row, M)) + (t & 1);
    if (t < 2) __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)src, nyq, 4, 0, 0);
}

// wk[e] = exp(+2 pi i (t + 256 e)/N), table values (e = 4: i wk[0], applied as a rotation)
FB_DEV void rq_ext(cf *v, int t, const cf *stg, const cf *wk)
{
    constexpr int M = RowQ::M;
#ifdef FB_R8_NOEXT  /* timing experiment only */
    for (int e = 0; e < 8; ++e) v[e] = cf_make(1.f + e, 1.f + t);
    return;
#endif
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = t + 256 * e;
        cf a = lds_rd(&stg[k]), b = lds_rd(&stg[M - k]);
        if (e == 0 && t == 0) { a.y = 0.f; b.y = 0.f; }                               // k = 0: X[0] and X[M] count as real (SURVEY note N2)
        const cf ev = cadd_conj(a, b);
        cf d = cmul(csub_conj(a, b), wk[e == 4 ? 0 : e]);                              // (X[k] - conj X[M-k]) e^{2 pi i k/N}
        if (e == 4) d = mul_w16<4, +1>(d);
        v[e] = cadd_ib(ev, d);
    }
}

// physical-space index of (thread t, register e)
FB_DEV int rq_phys(int t, int e)
{
    const int w = t >> 6, l = t & 63, l_hi = l >> 3, l_lo = l & 7;
    return (w + 4 * (l_hi >> 2)) + 8 * (l_hi & 3) + 32 * l_lo + 256 * e;
}

// vort_src [x][y] -> float2 (y = 2j, 2j+1) at [x][e][t], j = rq_phys(t, e)
__global__ void __launch_bounds__(256) k_rowq_permute_src(const float *__restrict__ in, float *__restrict__ out, int nrows)
{
    constexpr int M = RowQ::M;
    const size_t total = (size_t)nrows * M;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(idx / M), r = (int)(idx - (size_t)x * M);
        const int t = r & 255, e = r >> 8;
        reinterpret_cast<float2 *>(out)[idx] = reinterpret_cast<const float2 *>(in)[(size_t)x * M + rq_phys(t, e)];
    }
}

// PRE: the four fields arrive multiplied by 1/GRIDS (k_col_full, FullArgs::wscale): GRIDS is a power of two on this path, so the
// scaling commutes exactly with every rounding of the transforms and the 32 multiplications per thread and row are not needed here
// LOOP: a persistent grid, each workgroup loops over rows (FB_ROW_GRID experiments).  The default launch has one workgroup per row; with
// the trip count known to be one the compiler no longer hoists the store offsets out of the row loop (which cost a spilled register whose
// reload sat behind the first store: a wait for that store's acknowledgement in every workgroup).
template <bool SLAB, bool PRE = false, bool LOOP = false>
__global__ void __launch_bounds__(256, 4) k_rowq(RowArgs a, const float4 *__restrict__ tab /* per-thread twiddles, [9][256] float4: make_rowq_table() */)
{
    constexpr int M = RowQ::M;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf *xbuf = reinterpret_cast<cf *>(smem_raw);
    cf *stg = xbuf + RowQ::XBUF;
    const int t = threadIdx.x, l = t & 63;
    const int iters = LOOP ? (a.nx + gridDim.x - 1) / gridDim.x : 1;
#ifdef FB_ROW_SAMEROW   /* timing experiment only: every workgroup works on row 0 (no HBM traffic); results are wrong */
    auto row_of = [&](int it, bool &valid) { const int r = it * gridDim.x + blockIdx.x; valid = r < a.nx; return a.x0; };
#else
    auto row_of = [&](int it, bool &valid) { const int r = it * gridDim.x + blockIdx.x; valid = r < a.nx; return a.x0 + (valid ? r : 0); };
#endif
    // One workgroup per row: its start-up is on the critical path 4096 times per launch.  The first field's row is sent for BEFORE the
    // twiddle tables are read, so that the two round trips to memory overlap (vmcnt retires in order: the tables' arrival implies the row's).
    if (iters > 0) {
        bool vld; const int x = row_of(0, vld);
        rq_dma_issue<SLAB>(stg, t, a.M, 0, x);
    }
    // this thread's twiddles, one coalesced table (nine 16-byte loads; gathered from the root tables they were twelve loads that
    // touched up to 28 cache lines per wave instruction):  W_2048^{p t} (p = 1..7), W_256^{q l} (q = 1..3), W_4096^{t}, for t < 64 the LDS
    // table entry W_64^{p l_lo} at [p = t >> 3][l_lo = t & 7], and W_4096^{t + 256 e} for e = 1, 2, 3, 5, 6, 7 (the even/odd twiddle of every
    // element as ONE table value: it used to be W_4096^t times a 16th root, two more packed instructions per element and one more rounding)
    RowQTw tw;
    cf wkn[8];                                                                         // exp(+2 pi i (t + 256 e)/4096); wk[4] unused
    cf *tw2 = stg + RowQ::STG;
    {
        float4 q[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) q[j] = tab[j * 256 + t];
        tw.w0[0] = cf_make(q[0].x, q[0].y); tw.w0[1] = cf_make(q[0].z, q[0].w); tw.w0[2] = cf_make(q[1].x, q[1].y); tw.w0[3] = cf_make(q[1].z, q[1].w);
        tw.w0[4] = cf_make(q[2].x, q[2].y); tw.w0[5] = cf_make(q[2].z, q[2].w); tw.w0[6] = cf_make(q[3].x, q[3].y);
        tw.w1[0] = cf_make(q[3].z, q[3].w); tw.w1[1] = cf_make(q[4].x, q[4].y); tw.w1[2] = cf_make(q[4].z, q[4].w);
        wkn[0] = cf_make(q[5].x, -q[5].y);
        if (t < 64) tw2[t] = cf_make(q[5].z, q[5].w);
        wkn[1] = cf_make(q[6].x, -q[6].y); wkn[2] = cf_make(q[6].z, -q[6].w); wkn[3] = cf_make(q[7].x, -q[7].y);
        wkn[5] = cf_make(q[7].z, -q[7].w); wkn[6] = cf_make(q[8].x, -q[8].y); wkn[7] = cf_make(q[8].z, -q[8].w);
        wkn[4] = wkn[0];
    }
    tw.w2 = tw2;
#pragma unroll
    for (int p = 0; p < 7; ++p) asm volatile("" :: "v"(tw.w0[p]));                    // land the table loads here, not behind a later prefetch
    asm volatile("" :: "v"(tw.w1[0]), "v"(tw.w1[1]), "v"(tw.w1[2]), "v"(wkn[0]), "v"(wkn[7]));
    RH_WAIT_ROW();
    __syncthreads();

    for (int it = 0; it < iters; ++it) {
        bool valid;
        const int x = row_of(it, valid);                              // an invalid workgroup recomputes row 0, stores nothing
        bool vn = false;
        const int xn = (it + 1 < iters) ? row_of(it + 1, vn) : -1;
        cf v[8], p[8];
        auto c2r_phase = [&](bool wait, int next_field, int next_row) {
            const int tp = RQ_LAUNDER(t);
            if (wait) RH_WAIT_ROW();
            lds_barrier();
            rq_ext(v, tp, stg, wkn);
#ifndef RQ_LATE_DMA   /* -DRQ_LATE_DMA: the next row is sent for behind the exchange barrier instead (one barrier fewer per phase; measured: no gain at ny = 4096 and 16384, 2 % slower in k_rowh2) */
            lds_barrier();
            if (next_row >= 0) rq_dma_issue<SLAB>(stg, tp, a.M, next_field, next_row);
            rq_bwd(v, xbuf, tw, tp >> 6, tp & 63, [] {});
#else
            // the next row is sent for behind the barrier of the transform's own four-wave exchange: by then every wave has read
            // what it needs of the staged row
            rq_bwd(v, xbuf, tw, tp >> 6, tp & 63, [&] { if (next_row >= 0) rq_dma_issue<SLAB>(stg, tp, a.M, next_field, next_row); });
#endif
        };
        c2r_phase(false, 2, x);                                       // d vort/dx                         main.cpp:154
#pragma unroll
        for (int e = 0; e < 8; ++e) p[e] = PRE ? v[e] : cf_make(v[e].x * a.scale, v[e].y * a.scale);
        c2r_phase(true, 1, x);                                        // d psi/dy: -u * dvortdx = (c2r * scale) * dvortdx   main.cpp:200-201,225
#pragma unroll
        for (int e = 0; e < 8; ++e) p[e] = PRE ? cf_make(v[e].x * p[e].x, v[e].y * p[e].y) : cf_make((v[e].x * a.scale) * p[e].x, (v[e].y * a.scale) * p[e].y);
        c2r_phase(true, 3, x);                                        // d vort/dy                         main.cpp:168
        {
            cf zy[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) zy[e] = PRE ? v[e] : cf_make(v[e].x * a.scale, v[e].y * a.scale);
            c2r_phase(true, 0, xn);                                   // d psi/dx; the next row's first field travels meanwhile
#pragma unroll
            for (int e = 0; e < 8; ++e)                               // main.cpp:214,225-227
                v[e] = PRE ? cf_make(p[e].x - v[e].x * zy[e].x, p[e].y - v[e].y * zy[e].y)
                           : cf_make(p[e].x - (v[e].x * a.scale) * zy[e].x, p[e].y - (v[e].y * a.scale) * zy[e].y);
        }
        const int tt = RQ_LAUNDER2(t);
        if (a.src && a.src_nz[x]) {                                   // + vort_src (permuted order); rows of zeros are skipped
            const float2 *sp = reinterpret_cast<const float2 *>(a.src) + (size_t)x * M + tt;
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float2 q = sp[e * 256]; v[e].x += q.x; v[e].y += q.y; }
        }
        rq_fwd(v, xbuf, tw, tt >> 6, tt & 63);                        // main.cpp:237 (y part)
        RH_WAIT_ROW();                                                // the next row's first field has landed
        lds_barrier();
#pragma unroll
        for (int e = 4; e < 8; ++e) lds_wr(&xbuf[tt + 256 * (e - 4)], v[e]);          // W[k], k >= M/2, at k - M/2
        lds_barrier();
#ifdef FB_R8_NOST   /* timing experiment only: (almost) no stores */
        if (valid && v[0].x == 123.456f) {
#else
        if (valid) {
#endif
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = tt + 256 * e;                           // 0 <= k < M/2
                const cf wk = v[e];
                if (k == 0) {                                         // T[0] = Re W0 + Im W0 ; T[M] = Re W0 - Im W0
                    if (row_keep<SLAB>(a.T, a.t_frozen, 0)) *const_cast<cf *>(row_ptr<SLAB>(a.T, 0, x, 0)) = cf_make(wk.x + wk.y, 0.f);
                    if (row_keep<SLAB>(a.T, a.t_frozen, M)) *const_cast<cf *>(row_ptr<SLAB>(a.T, 0, x, M)) = cf_make(wk.x - wk.y, 0.f);
                    continue;
                }
                const cf wm = lds_rd(&xbuf[M / 2 - k]);
                const cf ev = cf_make(0.5f * (wk.x + wm.x), 0.5f * (wk.y - wm.y));
                const cf od = cf_make(0.5f * (wk.y + wm.y), 0.5f * (wm.x - wk.x));
                const cf co = cmulc(od, wkn[e]);                       // e^{-2 pi i k/N} O
                if (row_keep<SLAB>(a.T, a.t_frozen, k)) st2<(RQ_NT & 2) != 0>(const_cast<cf *>(row_ptr<SLAB>(a.T, 0, x, k)), cadd(ev, co));
                const cf tm = csub(ev, co);
                if (row_keep<SLAB>(a.T, a.t_frozen, M - k)) st2<(RQ_NT & 2) != 0>(const_cast<cf *>(row_ptr<SLAB>(a.T, 0, x, M - k)), cf_make(tm.x, -tm.y));
            }
            if (tt == 0 && row_keep<SLAB>(a.T, a.t_frozen, M / 2)) {
                const cf wh = v[4];
                *const_cast<cf *>(row_ptr<SLAB>(a.T, 0, x, M / 2)) = cf_make(wh.x, -wh.y);
            }
        }
    }
}
