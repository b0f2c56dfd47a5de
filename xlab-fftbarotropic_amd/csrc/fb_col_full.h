// fb_col_full.h -- single-pass x-transform + spectral update for nx = 4096 and nx = 8192 (one GPU); the default there.
//
// Replaces the three column kernels of a stage (k_col_strided<-1>, k_col_mid, k_col_strided<+1>:
// 17.5 C of measured traffic) by one kernel that moves 11.3-13.7 C: a 1024-thread workgroup keeps a whole
// tile of 8 ky-columns x 4096 x-rows (256 KiB) in its registers and runs the length-4096 transforms as
// 16 x 16 x 16 with one cross-wave and one in-wave LDS exchange per transform.  Same arithmetic as k_col_mid
// (main.cpp:148,179,198,212,237,240-251,296-312; fftwfop.cpp:87-124) with a different FFT factorisation.
// Measured at 4096^2: 0.163 ms per stage against 0.21 ms for the three kernels (DESIGN.md sections 4, 7).
//
//   thread (w = wave 0..15, l = (lane>>2) 0..15, c = lane&3) holds, for columns 2c,2c+1 of the tile:
//     mixed space   : rows x = 256 i + 16 w + l,            register index i  = 0..15
//     spectral space: kx   = k1 + 16 k2 + 256 k3, k1 = w, k2 = l, register index k3 = 0..15
//   forward : radix-16 over i -> k1 | x W256^{w k1} | wave<->reg transpose | radix-16 over w -> k2 |
//             x W4096^{l (k1+16 k2)} | lane<->reg transpose | radix-16 over l -> k3
//   backward: the same steps reversed with conjugated twiddles.
//
// Row segments in HBM are 64 B (8 columns); workgroups handling adjacent tiles are placed on the
// same XCD so that the two halves of every 128-B line meet in one L2 (tools/mb_strided.hip:
// 3.8-3.9 TB/s with that placement against 2.6-3.0 TB/s without).
//
// nx = 8192 = 2 x 4096 (NSUB = 2): workgroup (tile, k1) transforms the 4096 wavenumbers kx = 2 k2 + k1 of its tile; the
// remaining radix-2 step over x = x2 + 4096 x1 is fused into the row pass (fb_rowh.h, k_rowh<1, 2>), which reads the two
// "half-transformed" rows Y_k1[x2] of every field and writes the two rows U_k1[x2] of the tendency:
//     backward  X[x2 + 4096 x1] = Y_0[x2] + (-1)^x1 e^{+2 pi i x2/8192} Y_1[x2],   Y_k1 = IDFT_4096 over k2 of Z[2 k2 + k1]
//     forward   T[2 k2 + k1]   = DFT_4096 over x2 of U_k1[x2],   U_k1[x2] = e^{-2 pi i k1 x2/8192} (t[x2] + (-1)^k1 t[x2 + 4096])
// The mixed arrays then hold row k1*4096 + x2; nothing else changes (17.8 C of traffic per stage at 8192^2 instead of 22.4 C).
//
// The ky = ny/2 column lies outside the dealiasing circle on these grids (fftwfop.cpp:57-61): its tendency is always
// masked, its state never changes (SURVEY note N1).  The per-stage launches leave it alone, which makes their column count a
// power of two (ny/2 = 256 tiles of 8 at 4096^2: one tile per CU, no tail); the PRIME launch (STAGE = 4: derivatives of
// vort_c only, after the state was set) covers one more tile, [ny/2, ny/2 + 8) = that column plus zero padding, so its four
// derivative columns are written once.  Tiles that lie entirely outside the circle (ky >= 1936 at 4096^2) are skipped per
// stage for the same reason -- bit-identical, since the priming launch ran the same code on them.
//
// State arrays (ZA, ZB, ACC) use a layout private to this kernel:
//   [k1][tile, 0..ntiles][k3][thread] float4 = (column 2c, column 2c+1)   -- fully coalesced, 16 B per lane.
#pragma once
#include "fb_kernels.h"

#define CF_THREADS 1024
#define CF_X1_CF 16384                   /* cross-wave exchange: [k1][w][lane] complex             */
#define CF_X2_STR 68                     /* in-wave exchange: [reg][l][c] with 4 complex pad / reg */
#define CF_X2_CF (16 * 16 * CF_X2_STR)
#define CF_GX_CF 2048                    /* gradx_coe[4096] as floats                               */
#define CF_LDS_CF (CF_X2_CF + 512 + CF_GX_CF)  /* + tabB[256] (W256^m) + tabA[256] (W4096^m, m < 256) + gradx_coe */
#define CF_LDS_BYTES (CF_LDS_CF * 8)

struct FullArgs {
    const cf *Tin;        // tendency after the row pass, mixed layout [x][P]
    const cf *Zbase;      // vort_c0 (layout above)
    cf *Zcur, *Acc, *Zout;
    cf *W4;               // four derivative fields, mixed layout, field f at W4 + f*fstride
    long fstride;
    int P;                // pitch of the mixed arrays
    int ntiles;           // (ny/2)/8; the state arrays hold ntiles + 1 tiles per sub-sequence (the last one: the ky = ny/2 column)
    int ntiles_run;       // tiles this launch covers: PRIME ntiles + 1; stages: tiles that hold at least one column inside the dealiasing circle, the rest are frozen (note N1)
    int nsub;             // 1: nx = 4096; 2: nx = 8192 (kx = 2 k2 + k1, see above)
    int stage;            // 0..3; 4 = PRIME
    float nu, dt;
    SpecCoef coef;
    const cf *tw256;      // W256^m, m < 256
    const cf *tw4096;     // W4096^m (first 256 entries used)
    long sub_rows;        // rows of the mixed arrays per sub-sequence (4096)
    float wscale;         // the derivative fields leave multiplied by this: 1, or 1/GRIDS (a power of two: exact) when the row pass
                          // that reads them skips its own /GRIDS (k_rowq<false, true>; main.cpp:154-214 normalise after the c2r)
};

// 16x16 transpose between the wave index and the register index (one column = 8 B per lane)
FB_DEV void cf_xchg_waves(cf *lds, cf *v, int w, int lane)
{
#pragma unroll
    for (int r = 0; r < 16; ++r) lds_wr(&lds[(r * 16 + w) * 64 + lane], v[r]);
    lds_barrier();
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = lds_rd(&lds[(w * 16 + r) * 64 + lane]);
    lds_barrier();
}

// 16x16 transpose between l = lane>>2 and the register index, inside each wave's own LDS region.  The region
// is private (DS operations of a wave execute in order), so a workgroup barrier is needed only where the next
// LDS user is the cross-wave exchange, whose layout overlaps the other waves' regions (SYNC).
template <bool SYNC>
FB_DEV void cf_xchg_lanes(cf *lds, cf *v, int w, int l, int c)
{
    cf *reg = lds + w * (16 * CF_X2_STR);
#pragma unroll
    for (int r = 0; r < 16; ++r) lds_wr(&reg[r * CF_X2_STR + l * 4 + c], v[r]);
#ifdef CF_LANES_WAIT     /* not needed: the DS operations of a wave execute in order, the reads below see these writes (as in r8_xch_wave) */
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = lds_rd(&reg[l * CF_X2_STR + r * 4 + c]);
    if (SYNC) lds_barrier();                                 // the cross-wave exchange comes next
    else { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
}

// The two columns go through every step one after the other, fenced with sched_barrier: letting
// the scheduler interleave two radix-16 butterflies needs far more than the 128 VGPRs a
// 1024-thread workgroup has.
#define CF_FENCE() __builtin_amdgcn_sched_barrier(0)
// complex multiplies as two asm statements here (fb_fft_core.h, cmul_split): measured faster in this kernel
#ifndef CF_BFLY_SPLIT
#define CF_BFLY_SPLIT 1
#endif
#ifndef CF_TW_SPLIT
#define CF_TW_SPLIT 1
#endif
FB_DEV cf cf_mul(cf a, cf b) { return CF_TW_SPLIT ? cmul_split(a, b) : cmul(a, b); }
FB_DEV cf cf_mulc(cf a, cf b) { return CF_TW_SPLIT ? cmulc_split(a, b) : cmulc(a, b); }

// radix-16 butterfly with a scheduling fence after every radix-4 sub-butterfly: same arithmetic as
// Bfly<16>, but the four independent sub-butterflies are not interleaved (a few temporaries
// instead of four sets of them)
template <int DIR> FB_DEV void cf_bfly16(cf *v)
{
    fft4<DIR>(v[0], v[4], v[8], v[12]);  CF_FENCE();
    fft4<DIR>(v[1], v[5], v[9], v[13]);  CF_FENCE();
    fft4<DIR>(v[2], v[6], v[10], v[14]); CF_FENCE();
    fft4<DIR>(v[3], v[7], v[11], v[15]); CF_FENCE();
    v[5]  = mul_w16<1, DIR, CF_BFLY_SPLIT != 0>(v[5]);  v[6]  = mul_w16<2, DIR, CF_BFLY_SPLIT != 0>(v[6]);  v[7]  = mul_w16<3, DIR, CF_BFLY_SPLIT != 0>(v[7]);
    v[9]  = mul_w16<2, DIR, CF_BFLY_SPLIT != 0>(v[9]);  v[10] = mul_w16<4, DIR, CF_BFLY_SPLIT != 0>(v[10]); v[11] = mul_w16<6, DIR, CF_BFLY_SPLIT != 0>(v[11]);
    v[13] = mul_w16<3, DIR, CF_BFLY_SPLIT != 0>(v[13]); v[14] = mul_w16<6, DIR, CF_BFLY_SPLIT != 0>(v[14]); v[15] = mul_w16<9, DIR, CF_BFLY_SPLIT != 0>(v[15]);
    CF_FENCE();
    fft4<DIR>(v[0], v[1], v[2], v[3]);     CF_FENCE();
    fft4<DIR>(v[4], v[5], v[6], v[7]);     CF_FENCE();
    fft4<DIR>(v[8], v[9], v[10], v[11]);   CF_FENCE();
    fft4<DIR>(v[12], v[13], v[14], v[15]); CF_FENCE();
    cf t;
    t = v[1];  v[1]  = v[4];  v[4]  = t;
    t = v[2];  v[2]  = v[8];  v[8]  = t;
    t = v[3];  v[3]  = v[12]; v[12] = t;
    t = v[6];  v[6]  = v[9];  v[9]  = t;
    t = v[7];  v[7]  = v[13]; v[13] = t;
    t = v[11]; v[11] = v[14]; v[14] = t;
}

template <int DIR>
FB_DEV void cf_fft4096(cf *lds, const cf *tabA, const cf *tabB, cf (*v)[16], int w, int l, int c, int lane)
{
    if (DIR < 0) {
        // mixed -> spectral
#pragma unroll
        for (int col = 0; col < 2; ++col) {
            cf_bfly16<-1>(v[col]);                                       // over i -> k1
#pragma unroll
            for (int k1 = 1; k1 < 16; ++k1) { v[col][k1] = cf_mul(v[col][k1], tabB[w * k1]); if ((k1 & 3) == 3) CF_FENCE(); }
            cf_xchg_waves(lds, v[col], w, lane);                             // now wave = k1, reg = w
            CF_FENCE();
        }
        const cf a = tabA[l * w];                                            // W4096^{l k1}
#pragma unroll
        for (int col = 0; col < 2; ++col) {
            cf_bfly16<-1>(v[col]);                                       // over w -> k2
            v[col][0] = cf_mul(v[col][0], a);
#pragma unroll
            for (int k2 = 1; k2 < 16; ++k2) { v[col][k2] = cf_mul(v[col][k2], cf_mul(a, tabB[l * k2])); if ((k2 & 3) == 3) CF_FENCE(); }
            cf_xchg_lanes<false>(lds, v[col], w, l, c);                      // now l = k2, reg = l
            CF_FENCE();
        }
#pragma unroll
        for (int col = 0; col < 2; ++col) { cf_bfly16<-1>(v[col]); CF_FENCE(); }   // over l -> k3
    } else {
        // spectral -> mixed (thread: k1 = w, k2 = l, reg = k3)
#pragma unroll
        for (int col = 0; col < 2; ++col) {
            cf_bfly16<+1>(v[col]);                                       // over k3 -> l (in reg)
#pragma unroll
            for (int lr = 1; lr < 16; ++lr) { v[col][lr] = cf_mulc(v[col][lr], cf_mul(tabA[lr * w], tabB[lr * l])); if ((lr & 3) == 3) CF_FENCE(); }
            if (col == 0) cf_xchg_lanes<false>(lds, v[col], w, l, c);        // now lane-l = l, reg = k2
            else cf_xchg_lanes<true>(lds, v[col], w, l, c);
            CF_FENCE();
        }
#pragma unroll
        for (int col = 0; col < 2; ++col) {
            cf_bfly16<+1>(v[col]);                                       // over k2 -> w (in reg)
#pragma unroll
            for (int wr = 1; wr < 16; ++wr) { v[col][wr] = cf_mulc(v[col][wr], tabB[wr * w]); if ((wr & 3) == 3) CF_FENCE(); }
            cf_xchg_waves(lds, v[col], w, lane);                             // now wave = w, reg = k1
            CF_FENCE();
        }
#pragma unroll
        for (int col = 0; col < 2; ++col) { cf_bfly16<+1>(v[col]); CF_FENCE(); }   // over k1 -> i
    }
}

// Global accesses are written as (wave-uniform base pointer) + (32-bit per-lane byte offset): the
// compiler then uses the SGPR-base addressing form and one offset VGPR serves all 16 rows, instead
// of sixteen 64-bit per-lane addresses.
FB_DEV float4 cf_ld4(const void *ubase, unsigned voff) { return *reinterpret_cast<const float4 *>(static_cast<const char *>(ubase) + voff); }
FB_DEV void cf_st4(void *ubase, unsigned voff, float4 x) { *reinterpret_cast<float4 *>(static_cast<char *>(ubase) + voff) = x; }
// the derivative fields are read next by the row pass, a full sweep later: stream them (CF_NT_W4) so that they do
// not push this tile's freshly written state, which the next field re-reads, out of L2
#ifndef CF_NT_W4
#define CF_NT_W4 1     /* measured: 0.192 -> 0.166 ms per launch, and the row pass that follows 0.094 -> 0.087 ms */
#endif
#ifndef CF_NT_TIN      /* tendency rows (read once): -1.8 % per launch at 4096^2, -1.6 % at 8192^2 (no gain in round 1, before the other streams were sorted out) */
#define CF_NT_TIN 1
#endif
#ifndef CF_NT_Z0       /* vort_c0 (read once per stage): neutral to slightly worse */
#define CF_NT_Z0 0
#endif
#ifndef CF_NT_ZC       /* the previous stage state, read by stages 2 and 3 for the viscous term */
#define CF_NT_ZC 0
#endif
#ifndef CF_NT_RR3      /* the last of the three re-reads of the new state: -1 % alone, nothing on top of CF_NT_TIN */
#define CF_NT_RR3 0
#endif
#ifndef CF_NT_ACC      /* RK accumulator (next touched a whole stage later) */
#define CF_NT_ACC 1
#endif
FB_DEV void cf_st4_w4(void *ubase, unsigned voff, float4 x) { st4<CF_NT_W4 != 0>(static_cast<char *>(ubase) + voff, x); }

template <int STAGE, int NSUB>
__global__ void __launch_bounds__(CF_THREADS) k_col_full(FullArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf *lds = reinterpret_cast<cf *>(smem_raw);
    cf *tabB = lds + CF_X2_CF, *tabA = tabB + 256;
    // gradx_coe in LDS: vmcnt counts loads and stores in order, so a coefficient read from global memory
    // between two batches of stores would wait for the stores' acknowledgements (as k_col_mid used to)
    float *gxt = reinterpret_cast<float *>(tabA + 256);
    const int tid = threadIdx.x, lane = tid & 63, l = lane >> 2, c = lane & 3;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave id, in an SGPR

    // grid = nsub * ntiles_grid blocks; adjacent tiles on one XCD (blocks b and b+8 share an XCD: MI355X_MICROARCH.md, dispatch)
    constexpr int nsub = NSUB;
    const int ntg = gridDim.x / nsub;
    const int k1 = NSUB == 1 ? 0 : blockIdx.x / ntg, bt = blockIdx.x - k1 * ntg;
    int tile = bt;
    if ((ntg & 7) == 0) tile = (bt & 7) * (ntg >> 3) + (bt >> 3);
    if (tile >= a.ntiles_run) return;                     // frozen tile: state and derivatives stay what the priming launch left

    const int ky0 = tile * 8 + 2 * c;                                   // this thread's two columns: ky0, ky0+1
    const unsigned voff_m = (unsigned)((l * a.P + 2 * c) * (int)sizeof(cf));            // mixed arrays: row l of the wave's 16, column pair c
    const size_t ubase_m = ((size_t)(k1 * a.sub_rows + 16 * w) * a.P + (size_t)tile * 8) * sizeof(cf);    // uniform part: sub-sequence, rows 16 w.., tile's first column
    const size_t rstep = (size_t)256 * a.P * sizeof(cf);                                // 256 rows, bytes
    const unsigned voff_s = (unsigned)(lane * (int)sizeof(float4));                     // state arrays: [k1][tile][k3][thread]
    const size_t ubase_s = ((((size_t)k1 * (a.ntiles + 1) + tile) * 16) * CF_THREADS + (size_t)w * 64) * sizeof(float4);
    const size_t sstep = (size_t)CF_THREADS * sizeof(float4);

    constexpr bool PRIME = STAGE == 4;
    cf v[2][16];
    {   // ---- tendency tile -> registers (rows 256 i + 16 w + l); the tables are filled while these loads travel
        const char *src = reinterpret_cast<const char *>(a.Tin) + ubase_m;
        float4 tin[16];
        if (!PRIME) {
#pragma unroll
            for (int i = 0; i < 16; ++i) tin[i] = ld4<CF_NT_TIN != 0>(src + i * rstep + voff_m);
        }
        float g4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) g4[i] = a.coef.gx[nsub * (tid + i * CF_THREADS) + k1];      // gradx_coe at kx = nsub k2 + k1
        if (tid < 256) { tabB[tid] = a.tw256[tid]; tabA[tid] = a.tw4096[tid]; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {                   // kx2 = w + 16 l + 256 k3 sits at l + 16 w + 256 k3: the lanes of a wave (l) read neighbours
            const int ik = tid + i * CF_THREADS;
            gxt[((ik >> 4) & 15) + 16 * (ik & 15) + (ik & ~255)] = g4[i];
        }
        if (!PRIME) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { v[0][i] = cf_make(tin[i].x, tin[i].y); v[1][i] = cf_make(tin[i].z, tin[i].w); }
        }
    }
    __syncthreads();                                            // twiddle tables are in LDS
    if (!PRIME) cf_fft4096<-1>(lds, tabA, tabB, v, w, l, c, lane);           // main.cpp:237 (x part)

    // ---- viscous term, mask, RK stage update: kx = w + 16 l + 256 k3   (main.cpp:148,240-251,296-312)
    const char *Z0 = reinterpret_cast<const char *>(a.Zbase) + ubase_s;
    char *ZC = reinterpret_cast<char *>(a.Zcur) + ubase_s, *AC = reinterpret_cast<char *>(a.Acc) + ubase_s,
         *ZO = reinterpret_cast<char *>(a.Zout) + ubase_s;
    const float gya = a.coef.gy[ky0], gyb = a.coef.gy[ky0 + 1];
    // ky^2 is formed where it is used, (double)gy * (double)gy == the ky2 table entry (fftbaro.hip: h_ky2), from an opaque copy of gy:
    // held as doubles the two of them cost four registers for the whole tile (this kernel has none to spare)
    auto ky2_of = [](float g) { asm volatile("" : "+v"(g)); return (double)g * (double)g; };
    constexpr bool PRIME_ = STAGE == 4;
    const bool pada = PRIME_ && ky0 >= a.coef.hy, padb = PRIME_ && ky0 + 1 >= a.coef.hy;      // PRIME's last tile: columns beyond ny/2 are zero padding
    constexpr int stage = STAGE;
    const float nu = a.nu, dt = a.dt, hdt = (stage == 2) ? a.dt : a.dt / 2.0f;
    // The state arrays move one k3 at a time, CF_DEPTH k3 ahead: the loads of k3 + CF_DEPTH are issued before the stores of k3,
    // so no load is ever queued right behind a store (whose acknowledgement it would have to wait for).
    // Two k3 ahead where the registers allow it: stage 0 reads one array (0.140 -> 0.137 ms); with three arrays a depth of 2 spills
    // 100-250 B per lane and loses 3-6 %.
#ifndef CF_DEPTH
#define CF_DEPTH (STAGE == 0 ? 2 : 1)
#endif
    constexpr int DEPTH = CF_DEPTH;
    // Stage 1 does not read its stage state back: what stage 0 stored is fma(rk1, dt/2, vort_c0) with rk1 == the accumulator it also
    // stored, so the same instruction on the same bits gives it again (one array read less, 0.94 C of this launch's 12.6 C).
#ifndef CF_REMAKE_ZC
#define CF_REMAKE_ZC 1
#endif
    constexpr bool REMAKE_ZC = CF_REMAKE_ZC && STAGE == 1;
    float4 q0[DEPTH], q1[DEPTH], q2[DEPTH];
    auto load_k3 = [&](int k3) {
        const int sl = k3 % DEPTH;
        q0[sl] = ld4<CF_NT_Z0 != 0>(Z0 + k3 * sstep + voff_s);
        if (stage != 0) q2[sl] = ld4<CF_NT_ACC != 0>(AC + k3 * sstep + voff_s);
        if (stage != 0 && !REMAKE_ZC) q1[sl] = ld4<CF_NT_ZC != 0>(ZC + k3 * sstep + voff_s);
    };
    if (!PRIME) {
#pragma unroll
        for (int k3 = 0; k3 < DEPTH; ++k3) load_k3(k3);
    }
#pragma unroll
    for (int k3 = 0; k3 < (PRIME ? 0 : 16); ++k3) {
        const int sl = k3 % DEPTH;
        const int ik2 = w + 256 * k3 + 16 * l, ikx = nsub * ik2 + k1;
        const float gx = gxt[l + 16 * w + 256 * k3];
        const double kx2 = (double)gx * (double)gx;                      // fftwfop.cpp:42,45
        const float lapa = (float)(-(kx2 + ky2_of(gya))), lapb = (float)(-(kx2 + ky2_of(gyb)));
        const float mska = coef_mask(a.coef, ikx, ky0), mskb = coef_mask(a.coef, ikx, ky0 + 1);
        const float4 z0 = q0[sl];
        float4 zc = z0;
        if (REMAKE_ZC) {
            const float4 ac = q2[sl];
            zc = make_float4(__builtin_fmaf(ac.x, hdt, z0.x), __builtin_fmaf(ac.y, hdt, z0.y), __builtin_fmaf(ac.z, hdt, z0.z), __builtin_fmaf(ac.w, hdt, z0.w));
        } else if (stage != 0) zc = q1[sl];
        float4 k;
        k.x = (v[0][k3].x + (zc.x * lapa) * nu) * mska; k.y = (v[0][k3].y + (zc.y * lapa) * nu) * mska;
        k.z = (v[1][k3].x + (zc.z * lapb) * nu) * mskb; k.w = (v[1][k3].y + (zc.w * lapb) * nu) * mskb;
        float4 acc, zn;
        if (stage == 0) {
            acc = k;                   // (explicit fma: stage 1 recomputes this value from the stored accumulator, see REMAKE_ZC)
            zn = make_float4(__builtin_fmaf(k.x, hdt, z0.x), __builtin_fmaf(k.y, hdt, z0.y), __builtin_fmaf(k.z, hdt, z0.z), __builtin_fmaf(k.w, hdt, z0.w));
        } else if (stage < 3) {
            const float4 ac = q2[sl];
            acc = make_float4(ac.x + 2.0f * k.x, ac.y + 2.0f * k.y, ac.z + 2.0f * k.z, ac.w + 2.0f * k.w);
            zn = make_float4(z0.x + k.x * hdt, z0.y + k.y * hdt, z0.z + k.z * hdt, z0.w + k.w * hdt);
        } else {
            const float4 ac = q2[sl];
            acc = ac;
            zn = make_float4(z0.x + (ac.x + k.x) * dt / 6.0f, z0.y + (ac.y + k.y) * dt / 6.0f,
                             z0.z + (ac.z + k.z) * dt / 6.0f, z0.w + (ac.w + k.w) * dt / 6.0f);
        }
        v[0][k3] = cf_make(zn.x, zn.y); v[1][k3] = cf_make(zn.z, zn.w);
        if (k3 + DEPTH < 16) load_k3(k3 + DEPTH);
        if (stage < 3) { st4<CF_NT_ACC != 0>(AC + k3 * sstep + voff_s, acc); cf_st4(ZC + k3 * sstep, voff_s, zn); }
        else cf_st4(ZO + k3 * sstep, voff_s, zn);
        CF_FENCE();
    }

    // ---- four derivatives of the new state, each transformed back and stored (fftwfop.cpp:87-117)
    const char *ZN = PRIME ? Z0 : (stage < 3 ? ZC : ZO);
#pragma unroll 1
    for (int f = 0; f < 4; ++f) {
        if (f > 0 || PRIME) {                         // registers were consumed: fetch the state again (own writes, L2/MALL)
            const unsigned vs = (unsigned)launder((int)voff_s);
#pragma unroll
            for (int k3 = 0; k3 < 16; ++k3) {
                const float4 z = (CF_NT_RR3 && f == 3) ? ld4<true>(ZN + k3 * sstep + vs) : cf_ld4(ZN + k3 * sstep, vs);
                v[0][k3] = cf_make(z.x, z.y); v[1][k3] = cf_make(z.z, z.w);
            }
        }
        const bool psi = f >= 2, use_gx = (f == 0 || f == 3);
        const int lkx = launder(l);                               // per-iteration copy: keeps the table addresses inside the loop
#pragma unroll
        for (int k3 = 0; k3 < 16; ++k3) {
            cf za = v[0][k3], zb = v[1][k3];
            const float gx = gxt[16 * w + 256 * k3 + lkx];
            if (psi) {                                // psi_c = invertLaplacian(vort_c)   main.cpp:179
                const double kx2 = (double)gx * (double)gx;
                const float lia = (nsub * (w + 256 * k3 + 16 * l) + k1 == 0 && ky0 == 0) ? 1.0f : (float)(-(kx2 + ky2_of(gya)));
                const float lib = (float)(-(kx2 + ky2_of(gyb)));
                za = pada ? cf_make(0.f, 0.f) : cf_make(za.x / lia, za.y / lia);      // (-(kx^2 + 0) = 0 at kx = 0 in a padding column)
                zb = padb ? cf_make(0.f, 0.f) : cf_make(zb.x / lib, zb.y / lib);
            }
            const float ka = (use_gx ? gx : gya) * a.wscale, kb = (use_gx ? gx : gyb) * a.wscale;
            v[0][k3] = cf_make(-za.y * ka, za.x * ka);
            v[1][k3] = cf_make(-zb.y * kb, zb.x * kb);
            CF_FENCE();
        }
        cf_fft4096<+1>(lds, tabA, tabB, v, w, l, c, lane);
        char *dst = reinterpret_cast<char *>(a.W4) + (size_t)f * a.fstride * sizeof(cf) + ubase_m;
        const unsigned vm = (unsigned)launder((int)voff_m);
#pragma unroll
        for (int i = 0; i < 16; ++i)
            cf_st4_w4(dst + i * rstep, vm, make_float4(v[0][i].x, v[0][i].y, v[1][i].x, v[1][i].y));
    }
}

// ---- layout conversion: 3-pass private spectral layout (row N2*c+d holds kx = c + N1*d, pitch P) <-> this kernel's
// [k1][tile 0..ntiles][k3][thread][col]; columns beyond ny/2 (the last tile's padding) are zero
template <bool TO_FULL>
__global__ void __launch_bounds__(256) k_full_relayout(const cf *__restrict__ in, cf *__restrict__ out, int P, int N1, int N2, int ntiles, int nsub, int hy)
{
    // one thread per (k1, tile, k3, tid, col)
    const size_t total = (size_t)nsub * (ntiles + 1) * 16 * CF_THREADS * 2;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int col = (int)(idx & 1);
        const int tid = (int)((idx >> 1) & (CF_THREADS - 1));
        const int k3 = (int)((idx >> 11) & 15);
        const int st = (int)(idx >> 15);                             // k1 * (ntiles + 1) + tile
        const int k1 = st / (ntiles + 1), tile = st - k1 * (ntiles + 1);
        const int w = tid >> 6, l = (tid >> 2) & 15, c = tid & 3;
        const int kx = nsub * (w + 16 * l + 256 * k3) + k1, ky = tile * 8 + 2 * c + col;
        const int cc = kx % N1, d = kx / N1;                       // 3-pass layout: row N2*cc + d holds kx = cc + N1*d
        const size_t p3 = (size_t)(N2 * cc + d) * P + ky;
        if (ky >= hy) { if (TO_FULL) out[idx] = cf_make(0.f, 0.f); continue; }
        if (TO_FULL) out[idx] = in[p3]; else out[p3] = in[idx];
    }
}
