// fftbaro.hip -- C ABI (include/fftbaro.h) of the MI355X-native barotropic-vorticity engine.
// Host side: context/plan/tables, launch logic, model state machine.  Kernels: fb_kernels.h.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <set>
#include <string>
#include <utility>
#include <vector>

#include "../../include/fftbaro.h"
#include "fb_kernels.h"
#include "fb_col_full.h"
#include "fb_row3.h"
#include "fb_row8.h"
#include "fb_rowh.h"
#include "fb_rowq.h"

#ifndef FB_ROWQ_DEFAULT
#define FB_ROWQ_DEFAULT 1     /* ny = 4096 row pass: 1 = k_rowq (one row per 256-thread workgroup, four per CU), 0 = k_row8 (two rows per 512-thread workgroup) */
#endif

// --------------------------------------------------------------------------------------------
// errors
// --------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;
static int fail(int code, const std::string &msg) { g_last_error = msg; return code; }

#define HIPCHK(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(FB_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));         \
    } while (0)

extern "C" const char *fb_strerror(int s)
{
    switch (s) {
    case FB_OK: return "ok";
    case FB_EINVAL: return "invalid argument";
    case FB_ENOMEM: return "out of memory";
    case FB_EHIP: return "HIP runtime error";
    case FB_EIO: return "I/O error";
    case FB_EUNSUPPORTED: return "unsupported grid size";
    default: return "unknown status";
    }
}
extern "C" const char *fb_last_error(void) { return g_last_error.c_str(); }
extern "C" void fb_internal_set_error(const char *msg) { g_last_error = msg ? msg : ""; }   // fb_fieldio.cpp, fb_slab_comm.cpp
extern "C" int fb_version(void) { return 200; }
extern "C" int fb_device_count(int *count)
{
    if (!count) return fail(FB_EINVAL, "fb_device_count: NULL");
    *count = 0;
    HIPCHK(hipGetDeviceCount(count));
    return FB_OK;
}
extern "C" int fb_set_device(int ordinal) { HIPCHK(hipSetDevice(ordinal)); return FB_OK; }

static bool is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }
// powers of two 64..16384, or 3*2^k in 192..3072 (the reference's default NPTS = 768)
static bool size_ok(int n) { return (is_pow2(n) && n >= 64 && n <= 16384) || (n % 3 == 0 && is_pow2(n / 3) && n >= 192 && n <= 3072); }
extern "C" int fb_size_supported(int nx, int ny) { return size_ok(nx) && size_ok(ny); }

// --------------------------------------------------------------------------------------------
// context
// --------------------------------------------------------------------------------------------
// A set of local ky columns with a pitch of its own.  One GPU: one group holding every column.  Multi-GPU: groups
// 0 .. nact-1 = this rank's slabs of the ACTIVE columns (ky < world*KA: at least one mode inside the dealiasing circle, exchanged
// every RK stage; two groups where the stage is pipelined by column groups -- fb_slab_driver.h -- else one), the last group =
// its slab of the FROZEN columns beyond them (SURVEY note N1: their state never changes, so their derivative fields cross the
// links once, at priming).  A rank's active slab [rank*KA, (rank+1)*KA) is cut locally: group 0 holds its first columns, group 1
// the rest, so which rank owns which ky column does not depend on the cut.
struct ColGroup {
    int ncols;                  // local columns == pitch of every array of the group (multiple of 16)
    int ky0;                    // global ky of local column 0
    int nct_active;             // 16-column tiles that contain at least one unmasked ky
};

struct fb_ctx {
    int world, rank;            // slab decomposition: this process owns x rows [rank*XL, (rank+1)*XL)
    int XL;
    int ngroups, nact; ColGroup grp[3];   // nact active groups (1 or 2), then the frozen one if KF > 0
    int KA, KF, katot;          // world > 1: columns per rank of the active (all groups together) / frozen slabs, katot = world*KA
    int nx, ny, hy, P;          // P = grp[0].ncols: pitch of the private layouts on one GPU
    int N1, N2;                 // nx = N1*N2
    float lx, ly;
    hipStream_t stream;
    // device tables
    float *d_gx; double *d_kx2; float *d_gy; double *d_ky2; double gws;
    cf *d_tw_n1, *d_tw_n2, *d_tw_big, *d_tw_row_bwd, *d_tw_row_fwd, *d_tw_256;
    cf *d_tw_row3;              // W_ny^j for ny = 3*M (row pass = radix 3 x three length-M transforms) and ny = 4096 (k_row8), else NULL
    bool use_row8;              // fused row pass of ny = 4096 through k_row8 (FB_NO_ROW8=1: the Stockham kernel)
    int rowh_v;                 // fused row pass of ny = 8192 (1) / 16384 (2) through k_rowh (FB_NO_ROWH=1: 0 = the Stockham kernel)
    cf *d_tw_4096;              // W_4096^j for k_rowh's sub-transforms
    bool use_rowq;              // fused row pass of ny = 4096 through k_rowq (one real row per 4-wave workgroup) instead of k_row8
    cf *d_tw_2048;              // k_rowq's per-thread twiddle table (make_rowq_table)
    int pace_strided;           // pace the strided sub-pass's accesses (fields much larger than the caches)
    int col_chunks;             // x pass of a stage is issued in this many column chunks ...
    int col_streams;            // ... round-robin over this many streams, so that one chunk's kernel tails are filled by the next chunk
    hipStream_t aux[3];         // the extra streams (created on demand) and the fork/join events
    hipEvent_t ev_fork, ev_join[3];
    bool nyq_frozen;            // the ky = ny/2 column lies outside the dealiasing circle (always on square grids)
    cf *d_scratch;              // nx*P complex, lazily allocated (standalone r2c / c2r)
    // host copies of the 1-D tables (fb_get_tables)
    std::vector<float> h_gx, h_gy; std::vector<double> h_kx2, h_ky2;
    int max_wg;                 // grid cap (persistent-style grids)
    int dev;                    // HIP device the context lives on
};

static void split_nx(int nx, int &N1, int &N2)
{
    if (nx % 3 == 0) { N1 = 24; N2 = nx / 24; return; }      // 3*2^k: a 24-row strided sub-pass (radix 3 x 8)
    switch (nx) {
    case 64: N1 = 8; N2 = 8; break;
    case 128: N1 = 16; N2 = 8; break;
    case 256: N1 = 16; N2 = 16; break;
    case 512: N1 = 32; N2 = 16; break;
    case 1024: N1 = 32; N2 = 32; break;
    case 2048: N1 = 64; N2 = 32; break;
    case 4096: N1 = 64; N2 = 64; break;
    case 8192: N1 = 128; N2 = 64; break;
    default: N1 = 128; N2 = 128; break;   // 16384
    }
    // tuning override: FB_SPLIT_N1=<8|16|32|64|128> (both factors must stay within 8..128)
    if (const char *e = getenv("FB_SPLIT_N1")) {
        const int n1 = atoi(e);
        if (n1 >= 8 && n1 <= 128 && (n1 & (n1 - 1)) == 0 && nx % n1 == 0 && nx / n1 >= 8 && nx / n1 <= 128) { N1 = n1; N2 = nx / n1; }
    }
}

static std::vector<cf> make_root_table(int n)
{
    std::vector<cf> t(n);
    for (int j = 0; j < n; ++j) {
        double a = -2.0 * M_PI * (double)j / (double)n;
        t[j].x = (float)cos(a); t[j].y = (float)sin(a);
    }
    return t;
}

// k_rowq's per-thread twiddles, [9][256] float4 = eighteen complex per thread t (fb_rowq.h): W_2048^{p t} (p = 1..7), W_256^{q (t & 63)}
// (q = 1..3), W_4096^{t}, for t < 64 W_64^{(t >> 3)(t & 7)}, and W_4096^{t + 256 e} for e = 1, 2, 3, 5, 6, 7; the very values of the root tables
static std::vector<cf> make_rowq_table()
{
    const std::vector<cf> r2048 = make_root_table(2048), r4096 = make_root_table(4096);
    std::vector<cf> tab(18 * 256);
    for (int t = 0; t < 256; ++t) {
        cf e[18];
        for (int p = 1; p < 8; ++p) e[p - 1] = r2048[(p * t) % 2048];
        for (int q = 1; q < 4; ++q) e[6 + q] = r2048[(8 * q * (t & 63)) % 2048];
        e[10] = r4096[t];
        e[11] = t < 64 ? r2048[(32 * (t & 7) * (t >> 3)) % 2048] : cf{0.f, 0.f};
        const int es[6] = {1, 2, 3, 5, 6, 7};
        for (int i = 0; i < 6; ++i) e[12 + i] = r4096[t + 256 * es[i]];
        for (int j = 0; j < 9; ++j) { tab[(j * 256 + t) * 2] = e[2 * j]; tab[(j * 256 + t) * 2 + 1] = e[2 * j + 1]; }
    }
    return tab;
}

// Stockham stage tables for a radix list walked in order (forward-sign values); layout must
// match stockham_stage(): offset(s) + (m*(R-1) + (q-1))*T + t
static std::vector<cf> make_row_table(int n, const std::vector<int> &radices)
{
    const int T = n / 16;
    std::vector<cf> t;
    int NS = 1;
    for (size_t s = 0; s < radices.size(); ++s) {
        const int R = radices[s];
        if (s > 0) {
            const int NB = 16 / R;
            for (int m = 0; m < NB; ++m)
                for (int q = 1; q < R; ++q)
                    for (int tt = 0; tt < T; ++tt) {
                        const int j = tt + m * T, k = j % NS;
                        double a = -2.0 * M_PI * (double)k * (double)q / ((double)NS * (double)R);
                        cf w; w.x = (float)cos(a); w.y = (float)sin(a);
                        t.push_back(w);
                    }
        }
        NS *= R;
    }
    return t;
}

template <int N> static std::vector<int> plan_radices(bool fwd)
{
    std::vector<int> r;
    for (int s = 0; s < RowPlan<N>::S; ++s) r.push_back(RowTw<N, false>::radix(s));
    if (fwd) r = std::vector<int>(r.rbegin(), r.rend());
    return r;
}

static std::vector<int> plan_radices_rt(int n, bool fwd)
{
    switch (n) {
    case 64: return plan_radices<64>(fwd);
    case 128: return plan_radices<128>(fwd);
    case 256: return plan_radices<256>(fwd);
    case 512: return plan_radices<512>(fwd);
    case 1024: return plan_radices<1024>(fwd);
    case 2048: return plan_radices<2048>(fwd);
    case 4096: return plan_radices<4096>(fwd);
    case 8192: return plan_radices<8192>(fwd);
    default: return plan_radices<16384>(fwd);
    }
}

template <typename T> static int upload(T **dptr, const std::vector<T> &h)
{
    HIPCHK(hipMalloc((void **)dptr, h.size() * sizeof(T)));
    HIPCHK(hipMemcpy(*dptr, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return FB_OK;
}

static int autotune_pitch(fb_ctx *c);

extern "C" int fb_create(fb_ctx **out, int nx, int ny, float lx, float ly) { return fb_create_slab(out, nx, ny, lx, ly, 0, 1); }

static int round16(int v) { return (v + 15) / 16 * 16; }

// columns per rank of the active and frozen slabs (must match slab.py: slab_geometry)
static void slab_split(int ny, double gws, int world, int &jmax, int &KA, int &KF)
{
    const int hy = ny / 2 + 1;
    jmax = 0;                                              // first ky with ky^2 >= gws: that column and all beyond are masked (fftwfop.cpp:57-61)
    while (jmax < hy && (double)jmax * (double)jmax < gws) ++jmax;
    KA = round16((jmax + world - 1) / world);
    const int nf = hy - world * KA;
    KF = nf > 0 ? round16((nf + world - 1) / world) : 0;
}

// How many column groups the ACTIVE columns of a rank are cut into (fb_slab_driver.h: with two, a stage's forward x pass,
// update and backward x pass start on the first group while the second group's tendency is still on the links, and the first
// group's derivative fields leave while the second group is computed).  Worth it where one group's column work hides a
// collective's latency several times over: ~17.5 passes over nx*KA complex at ~5 TB/s.  FB_SLAB_COL_GROUPS=1|2 overrides.
static int slab_active_groups(int nx, int world, int KA)
{
    int na = 1;
    if (world > 1 && KA >= 32) {
        const double col_us = 17.5 * (double)nx * KA * 8.0 / 5e6;
        if (col_us >= 100.0) na = 2;
        if (const char *e = getenv("FB_SLAB_COL_GROUPS")) { const int v = atoi(e); if (v == 1 || v == 2) na = v; }
    }
    return na;
}

extern "C" int fb_create_slab(fb_ctx **out, int nx, int ny, float lx, float ly, int rank, int world)
{
    if (!out) return fail(FB_EINVAL, "fb_create: out is NULL");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world || !is_pow2(world) || (nx / world) < 2 || ((nx / world) & 1))
        return fail(FB_EINVAL, "fb_create_slab: world must be a power of two with an even nx/world >= 2, 0 <= rank < world");
    if (!fb_size_supported(nx, ny))
        return fail(FB_EUNSUPPORTED, "fb_create: nx, ny must be powers of two in [64, 16384] or 3*2^k in [192, 3072]");
    if (!(lx > 0.f) || !(ly > 0.f)) return fail(FB_EINVAL, "fb_create: Lx, Ly must be positive");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(FB_EHIP, "fb_create: no HIP device (this engine has no CPU fallback)");

    fb_ctx *c = new fb_ctx();                              // value-initialised: every pointer NULL, so fb_destroy() is safe on any error path below
    c->nx = nx; c->ny = ny; c->hy = ny / 2 + 1;
    c->world = world; c->rank = rank; c->XL = nx / world;
    split_nx(nx, c->N1, c->N2);
    c->lx = lx; c->ly = ly; c->stream = nullptr; c->d_scratch = nullptr;

    // ---- coefficient tables: fftwfop.cpp:5-79 ----
    const float TWOPI = (float)(acosf(-1.0f) * 2.0f);                  // fftwfop.hpp:7
    const int hx = nx / 2 + 1;
    const int dxw = (int)ceil(((double)(float)nx) / 3.0), dyw = (int)ceil(((double)(float)ny) / 3.0);   // :11-12
    c->gws = (double)(float)((double)dxw * dxw + (double)dyw * dyw);              // :57
    int jmax = 0;
    slab_split(ny, c->gws, world, jmax, c->KA, c->KF);
    int Ptot;                                              // table length: every global ky a local column can map to
    if (world == 1) {
        c->P = round16(c->hy);
        if (const char *e = getenv("FB_PITCH_EXTRA")) c->P += 16 * atoi(e);   // tuning hook (disables the autotuner below)
        c->ngroups = c->nact = 1; c->grp[0] = ColGroup{c->P, 0, 0};
        c->KA = c->P; c->KF = 0; c->katot = c->P;
        Ptot = c->P + 96;                                  // room for the pitch candidates of autotune_pitch()
    } else {
        c->katot = world * c->KA;
        c->nact = slab_active_groups(nx, world, c->KA);
        c->ngroups = c->nact + (c->KF > 0 ? 1 : 0);
        const int tiles = c->KA / 16;
        int off = 0;
        for (int g = 0; g < c->nact; ++g) {                // whole 16-column tiles, the first group takes the odd one
            const int nc = 16 * (tiles / c->nact + (g < tiles % c->nact ? 1 : 0));
            c->grp[g] = ColGroup{nc, rank * c->KA + off, 0};
            off += nc;
        }
        if (c->KF > 0) c->grp[c->nact] = ColGroup{c->KF, c->katot + rank * c->KF, 0};
        c->P = c->grp[0].ncols;
        Ptot = c->katot + world * c->KF + 16;
    }
    c->h_gx.assign(nx, 0.f); c->h_kx2.assign(nx, 0.0); c->h_gy.assign(Ptot, 0.f); c->h_ky2.assign(Ptot, 0.0);
    for (int i = 0; i < hx; ++i) c->h_gx[i] = TWOPI * ((float)i) / lx;            // :15-17
    for (int i = hx; i < nx; ++i) c->h_gx[i] = -c->h_gx[nx - i];                  // :18-20
    for (int j = 0; j < c->hy; ++j) c->h_gy[j] = TWOPI * ((float)j) / ly;         // :22-24
    for (int i = 0; i < nx; ++i) c->h_kx2[i] = (double)c->h_gx[i] * (double)c->h_gx[i];
    for (int j = 0; j < c->hy; ++j) c->h_ky2[j] = (double)c->h_gy[j] * (double)c->h_gy[j];

    c->nyq_frozen = ((double)(ny / 2) * (double)(ny / 2) >= c->gws);
    c->col_chunks = 1;
    c->pace_strided = 0;      // measured: only pays at the unlucky pitch 16*129; off by default (FB_PACE=1 to try)
    if (const char *e = getenv("FB_PACE")) c->pace_strided = atoi(e) != 0;
    if (const char *e = getenv("FB_COL_CHUNKS")) { int v = atoi(e); if (v >= 1 && v <= 16) c->col_chunks = v; }
    c->col_streams = 1;
    if (const char *e = getenv("FB_COL_STREAMS")) { int v = atoi(e); if (v >= 1 && v <= 4) c->col_streams = v; }
    if (c->col_streams > c->col_chunks) c->col_streams = c->col_chunks;
    int rc;
    if ((rc = upload(&c->d_gx, c->h_gx)) || (rc = upload(&c->d_kx2, c->h_kx2)) ||
        (rc = upload(&c->d_gy, c->h_gy)) || (rc = upload(&c->d_ky2, c->h_ky2)) ||
        (rc = upload(&c->d_tw_n1, make_root_table(c->N1))) || (rc = upload(&c->d_tw_n2, make_root_table(c->N2))) ||
        (rc = upload(&c->d_tw_big, make_root_table(nx))) || (rc = upload(&c->d_tw_256, make_root_table(256))) ||
        (rc = upload(&c->d_tw_row_bwd, make_row_table(ny % 3 ? ny : ny / 3, plan_radices_rt(ny % 3 ? ny : ny / 3, false)))) ||
        (rc = upload(&c->d_tw_row_fwd, make_row_table(ny % 3 ? ny : ny / 3, plan_radices_rt(ny % 3 ? ny : ny / 3, true))))) {
        fb_destroy(c); return rc;
    }
    if ((ny % 3 == 0 || ny >= 4096) && (rc = upload(&c->d_tw_row3, make_root_table(ny)))) { fb_destroy(c); return rc; }   // 4096: k_row8's W_ny^j; 8192, 16384: k_rowh's
    c->use_row8 = ny == 4096 && !getenv("FB_NO_ROW8");
    c->rowh_v = (ny == 8192 || ny == 16384) && !getenv("FB_NO_ROWH") ? ny / 8192 : 0;
    if (c->rowh_v && (rc = upload(&c->d_tw_4096, make_root_table(4096)))) { fb_destroy(c); return rc; }
    { const char *e = getenv("FB_ROWQ"); c->use_rowq = ny == 4096 && !getenv("FB_NO_ROW8") && (e ? e[0] != '0' : FB_ROWQ_DEFAULT); }
    if (c->use_rowq && (rc = upload(&c->d_tw_2048, make_rowq_table()))) { fb_destroy(c); return rc; }
    hipDeviceProp_t prop;
    if (hipGetDevice(&c->dev) != hipSuccess || hipGetDeviceProperties(&prop, c->dev) != hipSuccess) {
        fb_destroy(c); return fail(FB_EHIP, "fb_create: cannot query the device");
    }
    c->max_wg = prop.multiProcessorCount * 8;
    if ((rc = autotune_pitch(c))) { fb_destroy(c); return rc; }
    *out = c;
    return FB_OK;
}

extern "C" int fb_destroy(fb_ctx *c)
{
    if (!c) return FB_OK;
    void *tabs[] = {c->d_gx, c->d_kx2, c->d_gy, c->d_ky2, c->d_tw_n1, c->d_tw_n2, c->d_tw_big, c->d_tw_row_bwd, c->d_tw_row_fwd, c->d_tw_256, c->d_tw_row3, c->d_tw_4096, c->d_tw_2048};
    for (void *t : tabs) if (t) hipFree(t);
    if (c->d_scratch) hipFree(c->d_scratch);
    for (int i = 0; i < 3; ++i) { if (c->aux[i]) hipStreamDestroy(c->aux[i]); if (c->ev_join[i]) hipEventDestroy(c->ev_join[i]); }
    if (c->ev_fork) hipEventDestroy(c->ev_fork);
    delete c;
    return FB_OK;
}

extern "C" int fb_set_stream(fb_ctx *c, void *s) { if (!c) return fail(FB_EINVAL, "ctx NULL"); c->stream = (hipStream_t)s; return FB_OK; }
extern "C" int fb_synchronize(fb_ctx *c) { if (!c) return fail(FB_EINVAL, "ctx NULL"); HIPCHK(hipStreamSynchronize(c->stream)); return FB_OK; }

static SpecCoef make_coef(const fb_ctx *c)
{
    SpecCoef s; s.gx = c->d_gx; s.kx2 = c->d_kx2; s.gy = c->d_gy; s.ky2 = c->d_ky2; s.gws = c->gws; s.nx = c->nx; s.hy = c->hy;
    s.gws_i = c->gws >= 2147483647.0 ? 2147483647 : (int)ceil(c->gws);
    return s;
}

extern "C" int fb_get_tables(fb_ctx *c, float *gx, float *gy, float *lap, float *lapi, float *mask)
{
    if (!c) return fail(FB_EINVAL, "ctx NULL");
    const int nx = c->nx, hy = c->hy;
    if (gx) memcpy(gx, c->h_gx.data(), sizeof(float) * nx);
    if (gy) memcpy(gy, c->h_gy.data(), sizeof(float) * hy);
    for (int i = 0; i < nx; ++i)
        for (int j = 0; j < hy; ++j) {
            const size_t k = (size_t)i * hy + j;
            const float l = (float)(-(c->h_kx2[i] + c->h_ky2[j]));
            if (lap) lap[k] = l;
            if (lapi) lapi[k] = (i == 0 && j == 0) ? 1.0f : l;
            if (mask) {
                const int ii = i < nx - i ? i : nx - i;
                mask[k] = ((double)ii * ii + (double)j * j >= c->gws) ? 0.0f : 1.0f;
            }
        }
    return FB_OK;
}

// --------------------------------------------------------------------------------------------
// buffers
// --------------------------------------------------------------------------------------------
extern "C" int fb_malloc(void **p, size_t bytes)
{
    if (!p) return fail(FB_EINVAL, "fb_malloc: NULL");
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipErrorOutOfMemory) return fail(FB_ENOMEM, "hipMalloc: out of memory");
    HIPCHK(e);
    return FB_OK;
}
extern "C" int fb_free(void *p) { if (p) HIPCHK(hipFree(p)); return FB_OK; }
extern "C" int fb_memcpy_h2d(fb_ctx *c, void *d, const void *h, size_t n)
{
    if (!c || !d || !h) return fail(FB_EINVAL, "fb_memcpy_h2d: NULL");
    HIPCHK(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, c->stream)); HIPCHK(hipStreamSynchronize(c->stream)); return FB_OK;
}
extern "C" int fb_memcpy_d2h(fb_ctx *c, void *h, const void *d, size_t n)
{
    if (!c || !d || !h) return fail(FB_EINVAL, "fb_memcpy_d2h: NULL");
    HIPCHK(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, c->stream)); HIPCHK(hipStreamSynchronize(c->stream)); return FB_OK;
}
extern "C" int fb_memset0(fb_ctx *c, void *d, size_t n)
{
    if (!c || !d) return fail(FB_EINVAL, "fb_memset0: NULL");
    HIPCHK(hipMemsetAsync(d, 0, n, c->stream)); return FB_OK;
}

// ---- asynchronous record path: pinned host memory, streams, events (opaque void* handles) ----
extern "C" int fb_malloc_host(void **p, size_t bytes)
{
    if (!p) return fail(FB_EINVAL, "fb_malloc_host: NULL");
    hipError_t e = hipHostMalloc(p, bytes, hipHostMallocDefault);
    if (e == hipErrorOutOfMemory) return fail(FB_ENOMEM, "hipHostMalloc: out of memory");
    HIPCHK(e);
    return FB_OK;
}
extern "C" int fb_free_host(void *p) { if (p) HIPCHK(hipHostFree(p)); return FB_OK; }
extern "C" int fb_stream_create(void **stream)
{
    if (!stream) return fail(FB_EINVAL, "fb_stream_create: NULL");
    hipStream_t s;
    HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (void *)s;
    return FB_OK;
}
extern "C" int fb_stream_destroy(void *stream) { if (stream) HIPCHK(hipStreamDestroy((hipStream_t)stream)); return FB_OK; }
extern "C" int fb_stream_synchronize(void *stream) { HIPCHK(hipStreamSynchronize((hipStream_t)stream)); return FB_OK; }
extern "C" int fb_event_create(void **ev)
{
    if (!ev) return fail(FB_EINVAL, "fb_event_create: NULL");
    hipEvent_t e;
    HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    *ev = (void *)e;
    return FB_OK;
}
extern "C" int fb_event_create_timing(void **ev)
{
    if (!ev) return fail(FB_EINVAL, "fb_event_create_timing: NULL");
    hipEvent_t e;
    HIPCHK(hipEventCreate(&e));
    *ev = (void *)e;
    return FB_OK;
}
extern "C" int fb_event_elapsed_ms(void *start, void *stop, float *ms)
{
    if (!start || !stop || !ms) return fail(FB_EINVAL, "fb_event_elapsed_ms: NULL");
    HIPCHK(hipEventSynchronize((hipEvent_t)stop));
    HIPCHK(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return FB_OK;
}
extern "C" int fb_event_destroy(void *ev) { if (ev) HIPCHK(hipEventDestroy((hipEvent_t)ev)); return FB_OK; }
extern "C" int fb_event_record(void *ev, void *stream) { if (!ev) return fail(FB_EINVAL, "event NULL"); HIPCHK(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream)); return FB_OK; }
extern "C" int fb_stream_wait_event(void *stream, void *ev) { if (!ev) return fail(FB_EINVAL, "event NULL"); HIPCHK(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)ev, 0)); return FB_OK; }
extern "C" int fb_event_synchronize(void *ev) { if (!ev) return fail(FB_EINVAL, "event NULL"); HIPCHK(hipEventSynchronize((hipEvent_t)ev)); return FB_OK; }
extern "C" int fb_memcpy_d2h_async(void *stream, void *h_dst, const void *d_src, size_t bytes)
{
    if (!h_dst || !d_src) return fail(FB_EINVAL, "fb_memcpy_d2h_async: NULL");
    HIPCHK(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return FB_OK;
}
extern "C" int fb_memcpy_h2d_async(void *stream, void *d_dst, const void *h_src, size_t bytes)
{
    if (!d_dst || !h_src) return fail(FB_EINVAL, "fb_memcpy_h2d_async: NULL");
    HIPCHK(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return FB_OK;
}

// --------------------------------------------------------------------------------------------
// pointwise launches
// --------------------------------------------------------------------------------------------
static int grid_for(const fb_ctx *c, size_t n, int block = 256)
{
    size_t g = (n + block - 1) / block;
    if (g > (size_t)c->max_wg) g = c->max_wg;
    return (int)(g ? g : 1);
}

#define NEED_SINGLE(c) do { if ((c)->world != 1) return fail(FB_EINVAL, "natural-layout entry point on a slab context (world > 1)"); } while (0)

template <int OP> static int launch_op(fb_ctx *c, const float *in, float *out)
{
    if (!c || !in || !out) return fail(FB_EINVAL, "operator: NULL argument");
    NEED_SINGLE(c);
    const size_t total = (size_t)c->nx * c->hy;
    hipLaunchKernelGGL((k_spec_op<OP>), dim3(grid_for(c, total)), dim3(256), 0, c->stream, make_coef(c), (const cf *)in, (cf *)out, c->hy, total);
    HIPCHK(hipGetLastError());
    return FB_OK;
}
extern "C" int fb_gradx(fb_ctx *c, const float *i, float *o) { return launch_op<OP_GRADX>(c, i, o); }
extern "C" int fb_grady(fb_ctx *c, const float *i, float *o) { return launch_op<OP_GRADY>(c, i, o); }
extern "C" int fb_laplacian(fb_ctx *c, const float *i, float *o) { return launch_op<OP_LAP>(c, i, o); }
extern "C" int fb_invert_laplacian(fb_ctx *c, const float *i, float *o) { return launch_op<OP_INVLAP>(c, i, o); }
extern "C" int fb_dealiase(fb_ctx *c, const float *i, float *o) { return launch_op<OP_DEALIAS>(c, i, o); }

extern "C" int fb_backward_normalize(fb_ctx *c, float *d)
{
    if (!c || !d) return fail(FB_EINVAL, "fb_backward_normalize: NULL");
    const size_t n = (size_t)c->nx * c->ny;
    hipLaunchKernelGGL(k_scale_real, dim3(grid_for(c, n)), dim3(256), 0, c->stream, d, (float)(int)n, 1, n);
    HIPCHK(hipGetLastError()); return FB_OK;
}
extern "C" int fb_negate(fb_ctx *c, float *d)
{
    if (!c || !d) return fail(FB_EINVAL, "fb_negate: NULL");
    const size_t n = (size_t)c->nx * c->ny;
    hipLaunchKernelGGL(k_scale_real, dim3(grid_for(c, n)), dim3(256), 0, c->stream, d, -1.0f, 0, n);
    HIPCHK(hipGetLastError()); return FB_OK;
}
extern "C" int fb_jacobian(fb_ctx *c, const float *u, const float *v, const float *dx, const float *dy, const float *src, float *out)
{
    if (!c || !u || !v || !dx || !dy || !out) return fail(FB_EINVAL, "fb_jacobian: NULL");
    const size_t n = (size_t)c->nx * c->ny;
    hipLaunchKernelGGL(k_jacobian, dim3(grid_for(c, n)), dim3(256), 0, c->stream, u, v, dx, dy, src, out, n);
    HIPCHK(hipGetLastError()); return FB_OK;
}
extern "C" int fb_spec_axpy(fb_ctx *c, float *acc, const float *x, float a)
{
    if (!c || !acc || !x) return fail(FB_EINVAL, "fb_spec_axpy: NULL");
    const size_t n = 2 * (size_t)c->nx * c->hy;
    hipLaunchKernelGGL(k_spec_axpy, dim3(grid_for(c, n)), dim3(256), 0, c->stream, (const float *)acc, x, a, acc, n);
    HIPCHK(hipGetLastError()); return FB_OK;
}
extern "C" int fb_spec_evolve(fb_ctx *c, const float *base, const float *rk, float a, float *out)
{
    if (!c || !base || !rk || !out) return fail(FB_EINVAL, "fb_spec_evolve: NULL");
    const size_t n = 2 * (size_t)c->nx * c->hy;
    hipLaunchKernelGGL(k_spec_axpy, dim3(grid_for(c, n)), dim3(256), 0, c->stream, base, rk, a, out, n);
    HIPCHK(hipGetLastError()); return FB_OK;
}
extern "C" int fb_spec_rk4_combine(fb_ctx *c, const float *base, const float *k1, const float *k2, const float *k3,
                                   const float *k4, float dt, float *out)
{
    if (!c || !base || !k1 || !k2 || !k3 || !k4 || !out) return fail(FB_EINVAL, "fb_spec_rk4_combine: NULL");
    const size_t n = 2 * (size_t)c->nx * c->hy;
    hipLaunchKernelGGL(k_rk4_combine, dim3(grid_for(c, n)), dim3(256), 0, c->stream, base, k1, k2, k3, k4, dt, out, n);
    HIPCHK(hipGetLastError()); return FB_OK;
}

// --------------------------------------------------------------------------------------------
// FFT pass launchers
// --------------------------------------------------------------------------------------------
// hipFuncSetAttribute is per device: remember which (kernel, device) pairs are done
static int set_max_lds(const fb_ctx *c, const void *fn, size_t bytes)
{
    static std::mutex mu; static std::set<std::pair<const void *, int>> done;
    std::lock_guard<std::mutex> lk(mu);
    if (done.count({fn, c->dev})) return FB_OK;
    HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    done.insert({fn, c->dev});
    return FB_OK;
}

template <int N, int MODE> static int launch_row_t(fb_ctx *c, const RowArgs &a)
{
    using C = RowCfg<N>;
    const int npairs = a.nx / 2;
    constexpr int ppw = (C::PAIR2 && MODE == ROW_FUSED) ? C::G / 2 : C::G;      // row pairs per workgroup (fb_kernels.h, RowCfg)
    int grid = (npairs + ppw - 1) / ppw;
    int cap = c->max_wg / 2;                  // persistent-style grid: a few workgroups per CU, each loops over row pairs
    if (const char *e = getenv("FB_ROW_GRID")) { int v = atoi(e); if (v >= 64) cap = v; }
    if (grid > cap) grid = cap;
    const bool slab = c->world > 1;
    auto kern = slab ? k_row<N, MODE, true> : k_row<N, MODE, false>;
    static size_t lds_extra = 0;              // experiment hook: FB_ROW_LDS_EXTRA=<bytes> lowers the workgroups per CU
    if (const char *e = getenv("FB_ROW_LDS_EXTRA")) lds_extra = (size_t)atol(e);
    int rc = set_max_lds(c, (const void *)kern, C::LDS_BYTES + lds_extra);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(C::THREADS), C::LDS_BYTES + lds_extra, c->stream, a);
    HIPCHK(hipGetLastError());
    return FB_OK;
}

template <int M, int MODE> static int launch_row3_t(fb_ctx *c, const RowArgs &a)
{
    using C = Row3Cfg<M>;
    const int npairs = a.nx / 2;
    constexpr int ppw = (C::TWO && MODE == ROW_FUSED) ? 1 : C::GP;     // row pairs per workgroup (fb_row3.h)
    int grid = (npairs + ppw - 1) / ppw;
    if (grid > c->max_wg) grid = c->max_wg;
    const bool slab = c->world > 1;
    auto kern = slab ? k_row3<M, MODE, true> : k_row3<M, MODE, false>;
    int rc = set_max_lds(c, (const void *)kern, C::LDS_BYTES);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(C::THREADS), C::LDS_BYTES, c->stream, a, (const cf *)c->d_tw_row3);
    HIPCHK(hipGetLastError());
    return FB_OK;
}

static int launch_row8(fb_ctx *c, const RowArgs &a)
{
    const int npairs = a.nx / 2;
    int grid = npairs, cap = c->max_wg / 4;   // two resident workgroups per CU, each loops over row pairs
    if (const char *e = getenv("FB_ROW_GRID")) { int v = atoi(e); if (v >= 64) cap = v; }
    if (grid > cap) grid = cap;
    auto kern = c->world > 1 ? k_row8<true> : k_row8<false>;
    int rc = set_max_lds(c, (const void *)kern, Row8::LDS_BYTES);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), Row8::LDS_BYTES, c->stream, a, (const cf *)c->d_tw_row3);
    HIPCHK(hipGetLastError());
    return FB_OK;
}

template <int V> static int launch_rowh(fb_ctx *c, const RowArgs &a)
{
    int grid = a.nx, cap = c->max_wg / (4 * V);   // resident workgroups: two per CU at ny = 8192, one at 16384; each loops over rows
    if (const char *e = getenv("FB_ROW_GRID")) { int v = atoi(e); if (v >= 64) cap = v; }
    if (grid > cap) grid = cap;
    auto kern = c->world > 1 ? k_rowh<V, true> : k_rowh<V, false>;
    int rc = set_max_lds(c, (const void *)kern, RowH<V>::LDS_BYTES);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), RowH<V>::LDS_BYTES, c->stream, a, (const cf *)c->d_tw_4096, (const cf *)c->d_tw_row3);
    HIPCHK(hipGetLastError());
    return FB_OK;
}

static int launch_rowq(fb_ctx *c, const RowArgs &a)
{
    // one workgroup per row, four resident per CU: measured 0.076-0.078 ms per launch at 4096^2 against 0.081-0.084 with a persistent
    // grid of 1024 looping over rows (the dispatcher's refill keeps the four contexts of a CU out of step; the prologue is cheap)
    int grid = a.nx;
    if (const char *e = getenv("FB_ROW_GRID")) { int v = atoi(e); if (v >= 64 && v < grid) grid = v; }
    const bool loop = grid < a.nx;
    auto kern = c->world > 1 ? (loop ? k_rowq<true, false, true> : k_rowq<true, false, false>)
              : a.prescaled ? (loop ? k_rowq<false, true, true> : k_rowq<false, true, false>)
                            : (loop ? k_rowq<false, false, true> : k_rowq<false, false, false>);
    int rc = set_max_lds(c, (const void *)kern, RowQ::LDS_BYTES);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), RowQ::LDS_BYTES, c->stream, a, (const float4 *)c->d_tw_2048);
    HIPCHK(hipGetLastError());
    return FB_OK;
}

static int launch_rowh2(fb_ctx *c, const RowArgs &a)
{
#ifdef RH2_SPLIT_EXPERIMENT   /* timing experiment (fb_rowh.h, k_rowh2s): two 512-thread workgroups per x2; results are wrong */
    if (getenv("FB_ROWH2_SPLIT")) {
        int grid = 2 * a.nx, cap = c->max_wg / 4;              // two resident workgroups per CU; FB_ROW_GRID=16384: one workgroup per (x2, h)
        if (const char *e = getenv("FB_ROW_GRID")) { int v = atoi(e); if (v >= 64) cap = v; }
        if (grid > cap) grid = cap;
        grid &= ~15;
        const size_t lds = RowH<1>::LDS_BYTES + 448 * sizeof(cf);
        int rc = set_max_lds(c, (const void *)k_rowh2s, lds);
        if (rc) return rc;
        hipLaunchKernelGGL(k_rowh2s, dim3(grid), dim3(512), lds, c->stream, a, (const cf *)c->d_tw_4096, (const cf *)c->d_tw_row3, grid / 2);
        HIPCHK(hipGetLastError());
        return FB_OK;
    }
#endif
    int grid = a.nx, cap = c->max_wg / 8;         // one 1024-thread workgroup per CU, each loops over x2
    if (const char *e = getenv("FB_ROW_GRID")) { int v = atoi(e); if (v >= 64) cap = v; }
    if (grid > cap) grid = cap;
    int rc = set_max_lds(c, (const void *)k_rowh2, RowH2::LDS_BYTES);
    if (rc) return rc;
    hipLaunchKernelGGL(k_rowh2, dim3(grid), dim3(1024), RowH2::LDS_BYTES, c->stream, a, (const cf *)c->d_tw_4096, (const cf *)c->d_tw_row3);
    HIPCHK(hipGetLastError());
    return FB_OK;
}

template <int MODE> static int launch_row(fb_ctx *c, const RowArgs &a)
{
    if (a.nx <= 0) return FB_OK;
    if (MODE == ROW_FUSED && c->use_rowq) return launch_rowq(c, a);
    if (MODE == ROW_FUSED && c->use_row8 && a.nx >= 2) return launch_row8(c, a);
    if (MODE == ROW_FUSED && c->rowh_v == 1) return launch_rowh<1>(c, a);
    if (MODE == ROW_FUSED && c->rowh_v == 2) return launch_rowh<2>(c, a);
    switch (c->ny) {
    case 192: return launch_row3_t<64, MODE>(c, a);
    case 384: return launch_row3_t<128, MODE>(c, a);
    case 768: return launch_row3_t<256, MODE>(c, a);
    case 1536: return launch_row3_t<512, MODE>(c, a);
    case 3072: return launch_row3_t<1024, MODE>(c, a);
    case 64: return launch_row_t<64, MODE>(c, a);
    case 128: return launch_row_t<128, MODE>(c, a);
    case 256: return launch_row_t<256, MODE>(c, a);
    case 512: return launch_row_t<512, MODE>(c, a);
    case 1024: return launch_row_t<1024, MODE>(c, a);
    case 2048: return launch_row_t<2048, MODE>(c, a);
    case 4096: return launch_row_t<4096, MODE>(c, a);
    case 8192: return launch_row_t<8192, MODE>(c, a);
    case 16384: return launch_row_t<16384, MODE>(c, a);
    }
    return fail(FB_EUNSUPPORTED, "row pass: unsupported ny");
}

static int col_grid(const fb_ctx *c, long ntiles)
{
    long g = (ntiles + 3) / 4;
    if (g > c->max_wg) g = c->max_wg;
    return (int)(g ? g : 1);
}

// natural rows (one block) or, for the 4-field exchange buffer of a slab model, [dst][field][XL][ncols]
static RowMap rowmap_natural() { RowMap r; r.xl_shift = 31; r.xl_mask = 0x7fffffff; r.dstride = 0; r.xl = 0; return r; }
static int ilog2(int v) { int s = 0; while ((1 << s) < v) ++s; return s; }
static RowMap rowmap_w4(const fb_ctx *c, const ColGroup &G)
{
    if (c->world == 1) return rowmap_natural();
    RowMap r; r.xl_shift = ilog2(c->XL); r.xl_mask = c->XL - 1; r.dstride = 4L * c->XL * G.ncols;
    r.xl = is_pow2(c->XL) ? 0 : c->XL;
    return r;
}
static size_t grp_elems(const fb_ctx *c, const ColGroup &G) { return (size_t)c->nx * G.ncols; }
static long w4_fstride(const fb_ctx *c, const ColGroup &G) { return c->world == 1 ? (long)grp_elems(c, G) : (long)c->XL * G.ncols; }

template <int DIR> static int launch_col_strided(fb_ctx *c, const ColGroup &G, cf *data, int nfields, long fstride, RowMap rm = rowmap_natural(),
                                                 int ct0 = 0, int nct = -1)
{
    if (nct < 0) nct = G.ncols / 16 - ct0;
    if (nct <= 0) return FB_OK;
    ColArgs a; a.data = data; a.fstride = fstride; a.rm = rm; a.ct0 = ct0; a.nct = nct; a.nfields = nfields; a.P = G.ncols; a.N1 = c->N1; a.N2 = c->N2;
    a.pace = c->pace_strided;
    a.tw_n = c->d_tw_n1; a.tw_big = c->d_tw_big;
    const long ntiles = (long)nfields * c->N2 * nct;
    const dim3 g(col_grid(c, ntiles)), b(256);
    switch (c->N1) {
    case 24: hipLaunchKernelGGL((k_col_strided<24, DIR>), g, b, 0, c->stream, a); break;
    case 8: hipLaunchKernelGGL((k_col_strided<8, DIR>), g, b, 0, c->stream, a); break;
    case 16: hipLaunchKernelGGL((k_col_strided<16, DIR>), g, b, 0, c->stream, a); break;
    case 32: hipLaunchKernelGGL((k_col_strided<32, DIR>), g, b, 0, c->stream, a); break;
    case 64: hipLaunchKernelGGL((k_col_strided<64, DIR>), g, b, 0, c->stream, a); break;
    case 128: hipLaunchKernelGGL((k_col_strided<128, DIR>), g, b, 0, c->stream, a); break;
    default: return fail(FB_EUNSUPPORTED, "col strided: N1");
    }
    HIPCHK(hipGetLastError());
    return FB_OK;
}

template <int DIR> static int launch_col_block(fb_ctx *c, const ColGroup &G, cf *data, int nfields, long fstride)
{
    if (G.ncols == 0) return FB_OK;
    ColArgs a; a.data = data; a.fstride = fstride; a.rm = rowmap_natural(); a.ct0 = 0; a.nct = G.ncols / 16; a.nfields = nfields; a.P = G.ncols; a.N1 = c->N1; a.N2 = c->N2;
    a.pace = 0;
    a.tw_n = c->d_tw_n2; a.tw_big = c->d_tw_big;
    const long ntiles = (long)nfields * c->N1 * (G.ncols / 16);
    const dim3 g(col_grid(c, ntiles)), b(256);
    switch (c->N2) {
    case 8: hipLaunchKernelGGL((k_col_block<8, DIR>), g, b, 0, c->stream, a); break;
    case 16: hipLaunchKernelGGL((k_col_block<16, DIR>), g, b, 0, c->stream, a); break;
    case 32: hipLaunchKernelGGL((k_col_block<32, DIR>), g, b, 0, c->stream, a); break;
    case 64: hipLaunchKernelGGL((k_col_block<64, DIR>), g, b, 0, c->stream, a); break;
    case 128: hipLaunchKernelGGL((k_col_block<128, DIR>), g, b, 0, c->stream, a); break;
    default: return fail(FB_EUNSUPPORTED, "col block: N2");
    }
    HIPCHK(hipGetLastError());
    return FB_OK;
}

static int launch_col_mid(fb_ctx *c, const MidArgs &a0)
{
    if (a0.nct <= 0) return FB_OK;
    MidArgs a = a0;
    const long ntiles = (long)c->N1 * a.nct;
    // small launches of small tiles (grids up to 1024^2): one tile per workgroup, the derivative fields spread over its four waves
    // (fb_kernels.h, MidArgs::split): 256^2 9072 -> 9808 steps/s, 1024^2 6084 -> 6382.  Not for 64-row tiles (the slabs of a 4096^2
    // multi-GPU rank: 0.336 -> 0.383 ms per step) nor for 2048^2 (0.026 -> 0.038 ms per launch).  FB_MID_SPLIT=0|1 overrides
    a.split = (ntiles <= 2048 && c->N2 <= 32) ? 1 : 0;
    if (const char *e = getenv("FB_MID_SPLIT")) a.split = e[0] == '1';
    const dim3 g(a.split ? (unsigned)(ntiles < c->max_wg ? ntiles : c->max_wg) : (unsigned)col_grid(c, ntiles)), b(256);
    switch (c->N2) {
    case 8: hipLaunchKernelGGL((k_col_mid<8>), g, b, 0, c->stream, a); break;
    case 16: hipLaunchKernelGGL((k_col_mid<16>), g, b, 0, c->stream, a); break;
    case 32: hipLaunchKernelGGL((k_col_mid<32>), g, b, 0, c->stream, a); break;
    case 64: hipLaunchKernelGGL((k_col_mid<64>), g, b, 0, c->stream, a); break;
    case 128: hipLaunchKernelGGL((k_col_mid<128>), g, b, 0, c->stream, a); break;
    default: return fail(FB_EUNSUPPORTED, "col mid: N2");
    }
    HIPCHK(hipGetLastError());
    return FB_OK;
}

static size_t priv_elems(const fb_ctx *c) { return grp_elems(c, c->grp[0]); }   // one GPU: the one group

// tiles of every group that hold at least one column inside the dealiasing circle (the rest is frozen forever)
static void finish_groups(fb_ctx *c)
{
    int jmax = 0;                                          // first ky with ky^2 >= gws (fftwfop.cpp:57-61)
    while (jmax < c->hy && (double)jmax * (double)jmax < c->gws) ++jmax;
    for (int g = 0; g < c->ngroups; ++g) {
        ColGroup &G = c->grp[g];
        const int act = (jmax - G.ky0 + 15) / 16;
        G.nct_active = act < 0 ? 0 : (act > G.ncols / 16 ? G.ncols / 16 : act);
        if (getenv("FB_NO_COLUMN_SKIP")) G.nct_active = G.ncols / 16;
    }
}

// The strided x sub-pass reads rows N2*P*8 bytes apart; how well that stride spreads over the HBM channels
// depends on the pitch in a way that is specific to the memory controller's address hash (measured on MI355X,
// backward strided pass of four fields: 4096^2: P = 2064 0.0955 ms, 2080 0.084 ms; 8192^2: P = 4112 0.393 ms,
// 4128 0.586 ms).  So on large single-GPU grids the pitch is chosen by timing that pass for a few candidates.
// Results do not depend on the pitch (pad columns are zero and every pass is linear).
// 0: the model's x pass will be the three column kernels; 1 / 2: the single-pass k_col_full with nx = 4096 / 8192 (fb_col_full.h)
static int full_pass_nsub(const fb_ctx *c)
{
    const char *fp = getenv("FB_FULL_PASS");
    if (c->world != 1 || !c->nyq_frozen || ((c->ny / 2) % 8) != 0 || (fp && fp[0] == '0')) return 0;
    if (c->nx == 4096) return 1;
    if (c->nx == 8192 && c->rowh_v == 1) return 2;
    return 0;
}

static int autotune_pitch(fb_ctx *c)
{
    finish_groups(c);
    if (c->world != 1 || getenv("FB_PITCH_EXTRA") || getenv("FB_NO_PITCH_TUNE")) return FB_OK;
    if ((size_t)c->nx * c->P * sizeof(cf) < ((size_t)32 << 20)) return FB_OK;      // cache-resident grids: nothing to gain
    const int nsub = full_pass_nsub(c);
    const int P0 = c->P, NC = nsub ? 7 : 4;                                        // create_impl leaves room for P0 + 96
    constexpr int PITCH_HEADROOM = 96, RULE_EXTRA = 32;                            // the coefficient tables hold P0 + PITCH_HEADROOM columns
    static_assert(RULE_EXTRA <= PITCH_HEADROOM && 16 * (7 - 1) <= PITCH_HEADROOM, "pitch candidates must fit the table head-room of create_impl");
    if ((int)c->h_gy.size() < P0 + PITCH_HEADROOM) return fail(FB_EINVAL, "autotune_pitch: coefficient tables shorter than the pitch candidates");
    // The fixed rule below was measured on MI355X (gfx950): the pitch's effect comes from that part's memory-controller address
    // hash.  On any other device the interleaved probe decides instead.
    hipDeviceProp_t prop;
    int dev = 0;
    const bool is_gfx950 = hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) == 0;
    if (nsub && is_gfx950 && !getenv("FB_PITCH_TUNE")) {
        // single-pass x transform: a fixed rule.  tools/pitch_scan.sh on three boxes: round16(ny/2 + 1) is the slowest pitch at
        // 4096^2 (k_col_full 0.1525-0.1556 ms against 0.148-0.152 for every other candidate) and, with + 16, at 8192^2 (0.75 / 0.79 ms
        // against 0.65-0.69 for + 32, + 64, + 96 and 0.69-0.72 for + 48, + 80).  The probe below (FB_PITCH_TUNE=1) sees the same
        // ordering on average but not reliably in one process: its buffers land on other physical pages than the model's, which
        // moves a candidate by up to 5 % at 8192^2, more than what separates the good pitches.
        c->P = c->KA = c->katot = c->grp[0].ncols = P0 + RULE_EXTRA;
        finish_groups(c);
        return FB_OK;
    }
    const size_t maxe = (size_t)c->nx * (P0 + 16 * (NC - 1));
    // the probe is the pitch-sensitive kernel of the path the model will take: the backward strided sub-pass of one field, or
    // a stage-1 launch of k_col_full (tendency in, three state arrays, 64-byte row segments of four derivative fields out, rows
    // P*8 bytes apart) on zeroed buffers.  tools/pitch_scan.sh: the step time follows this probe; PRIME launches with two
    // repetitions per candidate (the first version) did not resolve the 2-4 % between pitches and often kept the worst one.
    const int nbuf = nsub ? 9 : 1;           // k_col_full: Tin, 4 x W4, Z0, Zc, Acc, Zout
    cf *buf = nullptr;
    if (hipMalloc((void **)&buf, maxe * nbuf * sizeof(cf)) != hipSuccess) { hipGetLastError(); return FB_OK; }
    hipMemsetAsync(buf, 0, maxe * nbuf * sizeof(cf), c->stream);
    if (nsub) {
        const void *fn = nsub == 1 ? (const void *)k_col_full<1, 1> : (const void *)k_col_full<1, 2>;
        if (set_max_lds(c, fn, CF_LDS_BYTES)) { hipFree(buf); return FB_OK; }
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto probe = [&](int P) -> int {
        ColGroup G = c->grp[0]; G.ncols = P;
        if (!nsub) return launch_col_strided<+1>(c, G, buf, 1, 0);
        FullArgs a; memset(&a, 0, sizeof(a));
        a.Tin = buf; a.W4 = buf + maxe; a.Zbase = buf + 5 * maxe; a.Zcur = buf + 6 * maxe; a.Acc = buf + 7 * maxe; a.Zout = buf + 8 * maxe;
        a.fstride = (long)c->nx * P; a.P = P; a.ntiles = (c->ny / 2) / 8;
        a.ntiles_run = a.ntiles; a.nsub = nsub; a.stage = 1; a.nu = 0.f; a.dt = 0.f; a.coef = make_coef(c); a.tw256 = c->d_tw_256;
        a.tw4096 = nsub == 1 ? c->d_tw_big : c->d_tw_4096; a.sub_rows = 4096;
        const dim3 g(nsub * a.ntiles), b(CF_THREADS);
        if (nsub == 1) hipLaunchKernelGGL((k_col_full<1, 1>), g, b, CF_LDS_BYTES, c->stream, a);
        else hipLaunchKernelGGL((k_col_full<1, 2>), g, b, CF_LDS_BYTES, c->stream, a);
        return hipGetLastError() == hipSuccess ? FB_OK : FB_EHIP;
    };
    const bool verbose = getenv("FB_TUNE_VERBOSE") != nullptr;
    // the device needs ~25 ms of this load to reach full speed: run the probe that long before timing it
    hipEventRecord(e0, c->stream);
    for (int it = 0; it < 400; ++it) {
        if (probe(P0) != FB_OK) break;
        if ((it & 7) == 7) {
            hipEventRecord(e1, c->stream); hipEventSynchronize(e1);
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess || ms > 30.f) break;
        }
    }
    float tbest[8]; for (int k = 0; k < NC; ++k) tbest[k] = 1e30f;
    for (int rep = 0; rep < 6; ++rep)            // candidates interleaved, so that a drift of the clock hits all of them alike
        for (int k = 0; k < NC; ++k) {
            hipEventRecord(e0, c->stream);
            const int rc = probe(P0 + 16 * k);
            hipEventRecord(e1, c->stream);
            hipEventSynchronize(e1);
            float ms = 0.f;
            if (rc == FB_OK && hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms < tbest[k]) tbest[k] = ms;
        }
    float best = 1e30f; int bestP = P0;
    for (int k = 0; k < NC; ++k) {
        if (verbose) fprintf(stderr, "fftbaro: pitch probe P=%d %.4f ms\n", P0 + 16 * k, tbest[k]);
        if (tbest[k] < best * 0.99f) { best = tbest[k]; bestP = P0 + 16 * k; }     // a larger pitch must win by 1 % to be taken
    }
    if (verbose) fprintf(stderr, "fftbaro: pitch %d -> %d\n", P0, bestP);
    c->P = c->KA = c->katot = c->grp[0].ncols = bestP;
    finish_groups(c);
    hipEventDestroy(e0); hipEventDestroy(e1);
    hipFree(buf);
    hipGetLastError();
    return FB_OK;
}

static int ensure_scratch(fb_ctx *c)
{
    if (!c->d_scratch) {
        hipError_t e = hipMalloc((void **)&c->d_scratch, priv_elems(c) * sizeof(cf));
        if (e != hipSuccess) return fail(FB_ENOMEM, "scratch allocation failed");
    }
    return FB_OK;
}

// ---- row-pass views (fb_kernels.h: RowView) ----
static unsigned magic_div(int d) { return d > 0 ? (unsigned)((1ull << 32) / (unsigned)d) + 1u : 0u; }
// one GPU: nfields arrays of pitch P, fstride apart
static RowView view_single(const fb_ctx *c, const cf *base, long fstride)
{
    RowView v; memset(&v, 0, sizeof(v));
    v.a = v.b = v.f = base; v.fstrA = v.fstrB = v.fstrF = fstride; v.ka = v.ka0 = c->P; v.kb = 0; v.kf = 16; v.katot = 0x7fffffff;
    return v;
}
// multi-GPU exchange buffers, one per column group: [peer][nfields][XL][ncols_g]; bufs[g] = group g's (fb_ctx::grp order)
static RowView view_slab(const fb_ctx *c, const cf *const *bufs, int nfields)
{
    RowView v; memset(&v, 0, sizeof(v));
    v.a = bufs[0]; v.b = c->nact > 1 ? bufs[1] : bufs[0]; v.f = c->KF > 0 ? bufs[c->nact] : bufs[0];
    v.ka = c->KA; v.ka0 = c->grp[0].ncols; v.kb = c->nact > 1 ? c->grp[1].ncols : 0; v.kf = c->KF > 0 ? c->KF : 16;
    v.katot = c->katot;
    v.fstrA = (long)c->XL * v.ka0; v.fstrB = (long)c->XL * v.kb; v.fstrF = (long)c->XL * v.kf;
    v.sstrA = (long)nfields * v.fstrA; v.sstrB = (long)nfields * v.fstrB; v.sstrF = (long)nfields * v.fstrF;
    v.magA = magic_div(v.ka); v.magF = magic_div(v.kf);
    return v;
}
// the row-side view of one kind of exchange buffer of the model (which: 0 w4_recv, 1 t_send)
static RowArgs row_args_base(const fb_ctx *c)
{
    RowArgs a; memset(&a, 0, sizeof(a));
    a.x0 = 0; a.nx = c->XL; a.tw_bwd = c->d_tw_row_bwd; a.tw_fwd = c->d_tw_row_fwd; a.t_frozen = 1;
    return a;
}

// real [nx][ny] -> private spectral layout (dst: nx*P complex, pad columns must be zero-initialised)      [one GPU]
static int r2c_private(fb_ctx *c, const float *d_real, cf *dst)
{
    RowArgs a = row_args_base(c); a.rin = d_real; a.T = view_single(c, dst, 0);
    int rc;
    if ((rc = launch_row<ROW_FWD>(c, a))) return rc;
    if ((rc = launch_col_strided<-1>(c, c->grp[0], dst, 1, 0))) return rc;
    return launch_col_block<-1>(c, c->grp[0], dst, 1, 0);
}

// private spectral layout (work: destroyed) -> real [nx][ny] * scale                                       [one GPU]
static int c2r_private(fb_ctx *c, cf *work, float *d_real, float scale)
{
    int rc;
    if ((rc = launch_col_block<+1>(c, c->grp[0], work, 1, 0))) return rc;
    if ((rc = launch_col_strided<+1>(c, c->grp[0], work, 1, 0))) return rc;
    RowArgs a = row_args_base(c); a.M = view_single(c, work, 0); a.rout = d_real; a.scale = scale;
    return launch_row<ROW_INV>(c, a);
}

// 3-pass row layout <-> the tile-major layout of the state arrays (out-of-place)
static bool state_tm(const fb_ctx *c) { return c->N2 >= 32; }
static int state_convert(fb_ctx *c, const ColGroup &G, const cf *in, cf *out, bool to_tm)
{
    if (G.ncols == 0) return FB_OK;
    if (!state_tm(c)) { HIPCHK(hipMemcpyAsync(out, in, grp_elems(c, G) * sizeof(cf), hipMemcpyDeviceToDevice, c->stream)); return FB_OK; }
    const size_t total = grp_elems(c, G);
    if (to_tm) hipLaunchKernelGGL((k_state_relayout<true>), dim3(grid_for(c, total)), dim3(256), 0, c->stream, in, out, c->nx, G.ncols, c->N2);
    else hipLaunchKernelGGL((k_state_relayout<false>), dim3(grid_for(c, total)), dim3(256), 0, c->stream, in, out, c->nx, G.ncols, c->N2);
    HIPCHK(hipGetLastError());
    return FB_OK;
}

static int relayout(fb_ctx *c, const cf *in, cf *out, bool to_private)
{
    const size_t total = priv_elems(c);
    if (to_private) hipLaunchKernelGGL((k_spec_relayout<true>), dim3(grid_for(c, total)), dim3(256), 0, c->stream, in, out, c->nx, c->hy, c->P, c->N1, c->N2);
    else hipLaunchKernelGGL((k_spec_relayout<false>), dim3(grid_for(c, total)), dim3(256), 0, c->stream, in, out, c->nx, c->hy, c->P, c->N1, c->N2);
    HIPCHK(hipGetLastError());
    return FB_OK;
}

extern "C" int fb_r2c(fb_ctx *c, const float *d_in, float *d_out)
{
    if (!c || !d_in || !d_out) return fail(FB_EINVAL, "fb_r2c: NULL");
    NEED_SINGLE(c);
    int rc;
    if ((rc = ensure_scratch(c))) return rc;
    HIPCHK(hipMemsetAsync(c->d_scratch, 0, priv_elems(c) * sizeof(cf), c->stream));
    if ((rc = r2c_private(c, d_in, c->d_scratch))) return rc;
    return relayout(c, c->d_scratch, (cf *)d_out, false);
}

extern "C" int fb_c2r(fb_ctx *c, const float *d_in, float *d_out, int normalize)
{
    if (!c || !d_in || !d_out) return fail(FB_EINVAL, "fb_c2r: NULL");
    NEED_SINGLE(c);
    int rc;
    if ((rc = ensure_scratch(c))) return rc;
    if ((rc = relayout(c, (const cf *)d_in, c->d_scratch, true))) return rc;
    const float scale = normalize ? 1.0f / (float)((size_t)c->nx * c->ny) : 1.0f;
    return c2r_private(c, c->d_scratch, d_out, scale);
}

// --------------------------------------------------------------------------------------------
// fused RK4 model
// --------------------------------------------------------------------------------------------
// per column group: state arrays in the private layouts (nx*ncols complex each) and the exchange buffers.
// One GPU: w4_recv == w4_send (4 fields) and t_recv == t_send (no exchange).  Multi-GPU (engine-owned):
//   w4_send [dst][4][XL][ncols]  destination-blocked, written by the column kernels     -> all-to-all ->
//   w4_recv [src][4][XL][ncols]  block s = rank s's ky slab of this rank's rows, read by the row pass
//   t_send  [dst][XL][ncols]     written by the row pass                                -> all-to-all ->
//   t_recv  [nx][ncols]          block s = rows of rank s, read by the column kernels
struct GroupBufs { cf *ZA, *ZB, *ACC, *w4_send, *w4_recv, *t_send, *t_recv; };

struct fb_model {
    fb_ctx *c;
    float nu, dt;
    GroupBufs gb[3];
    bool phase_flow;                 // driven phase by phase (fb_slab_*): always the three-kernel column path
    // single-pass x-transform path (fb_col_full.h): ZA/ZB/ACC then use that kernel's private layout; nsub = nx/4096
    // (2: the remaining radix-2 step of the x transform is fused into the row pass, k_rowh2)
    bool full;
    int nsub;
    bool prescale;                   // k_col_full writes the derivative fields times 1/GRIDS and k_rowq<false, true> does not normalise (4096^2 on one GPU)
    // hipGraph replay of one RK4 step (launch-bound small grids): captured lazily on a non-null stream,
    // dropped whenever something baked into the kernel arguments changes (source pointer, stream)
    bool use_graph, warmed;
    hipGraphExec_t graph_exec;
    const float *graph_src; hipStream_t graph_stream;
    float *src;                      // vort_src (this rank's rows) or NULL (== zeros)
    int *src_nz;                     // per row of src: 1 = holds a non-zero value (k_src_row_flags)
    cf *nat[3];                      // natural-layout temporaries for the record path (lazy)
    // 0: derivative fields stale; 1: w4_send holds the derivatives with the backward x pass finished on the frozen
    // tiles only (the backward strided pass on the active tiles comes next); 2: finished on every tile (ready for the row pass)
    int primed;
};

static int model_create_impl(fb_model **out, fb_ctx *c, float nu, float dt, bool phase_flow)
{
    if (!out || !c) return fail(FB_EINVAL, "fb_model_create: NULL");
    *out = nullptr;
    fb_model *m = new fb_model();
    memset(m, 0, sizeof(*m));
    m->c = c; m->nu = nu; m->dt = dt; m->phase_flow = phase_flow;
    // single-pass x transform (fb_col_full.h) where it applies: one GPU, nx = 4096, frozen Nyquist column, whole
    // 8-column tiles.  0.177 ms per stage against 0.21 ms for the three column kernels; FB_FULL_PASS=0 keeps the latter.
    const char *fp = getenv("FB_FULL_PASS");
    (void)fp;
    m->nsub = phase_flow ? 0 : full_pass_nsub(c);
    m->full = m->nsub != 0;
    m->prescale = m->full && c->world == 1 && c->use_rowq && !getenv("FB_NO_PRESCALE");      // use_rowq: ny == 4096, so GRIDS = nx * ny is a power of two
    int rc = FB_OK;
    auto alloc0 = [&](cf **p, size_t elems) {              // zero-initialised device array (pad columns stay zero: every pass is linear)
        if (rc || elems == 0) return;
        if (hipMalloc((void **)p, elems * sizeof(cf)) != hipSuccess) { rc = fail(FB_ENOMEM, "model allocation failed"); return; }
        if (hipMemsetAsync(*p, 0, elems * sizeof(cf), c->stream) != hipSuccess) rc = fail(FB_EHIP, "hipMemsetAsync failed");
    };
    if (m->full) {
        const void *fns[10] = {(const void *)k_col_full<0, 1>, (const void *)k_col_full<1, 1>, (const void *)k_col_full<2, 1>, (const void *)k_col_full<3, 1>,
                               (const void *)k_col_full<4, 1>, (const void *)k_col_full<0, 2>, (const void *)k_col_full<1, 2>, (const void *)k_col_full<2, 2>,
                               (const void *)k_col_full<3, 2>, (const void *)k_col_full<4, 2>};
        for (int st = 0; st < 10 && !rc; ++st) rc = set_max_lds(c, fns[st], CF_LDS_BYTES);
    }
    for (int g = 0; g < c->ngroups; ++g) {
        const size_t n = grp_elems(c, c->grp[g]);
        GroupBufs &B = m->gb[g];
        alloc0(&B.ZA, n);
        if (g < c->nact) { alloc0(&B.ZB, n); alloc0(&B.ACC, n); }     // the frozen group's state never changes: vort_c only
        alloc0(&B.w4_send, 4 * n); alloc0(&B.t_send, n);
        if (c->world > 1) { alloc0(&B.w4_recv, 4 * n); alloc0(&B.t_recv, n); }
        else { B.w4_recv = B.w4_send; B.t_recv = B.t_send; }
    }
    if (rc) { fb_model_destroy(m); return rc; }
    *out = m;
    return FB_OK;
}

extern "C" int fb_model_create(fb_model **out, fb_ctx *c, float nu, float dt)
{
    if (c && c->world != 1) return fail(FB_EINVAL, "fb_model_create on a slab context: use fb_slab_create");
    return model_create_impl(out, c, nu, dt, false);
}

static void model_drop_graph(fb_model *m);

extern "C" int fb_model_destroy(fb_model *m)
{
    if (!m) return FB_OK;
    for (GroupBufs &B : m->gb) {
        if (B.w4_recv && B.w4_recv != B.w4_send) hipFree(B.w4_recv);
        if (B.t_recv && B.t_recv != B.t_send) hipFree(B.t_recv);
        cf *arr[] = {B.ZA, B.ZB, B.ACC, B.w4_send, B.t_send};
        for (cf *p : arr) if (p) hipFree(p);
    }
    model_drop_graph(m);
    if (m->src) hipFree(m->src);
    if (m->src_nz) hipFree(m->src_nz);
    for (auto p : m->nat) if (p) hipFree(p);
    delete m;
    return FB_OK;
}

extern "C" int fb_model_info(fb_model *m, size_t *hbm, size_t *alg)
{
    if (!m) return fail(FB_EINVAL, "model NULL");
    const fb_ctx *c = m->c;
    if (hbm) {
        size_t n = m->src ? (size_t)c->XL * c->ny * 4 : 0;
        for (int g = 0; g < c->ngroups; ++g) n += (c->world > 1 ? (g < c->nact ? 13 : 11) : 8) * grp_elems(c, c->grp[g]) * sizeof(cf);
        *hbm = n;
    }
    if (alg) *alg = (size_t)320 * c->nx * c->ny;
    return FB_OK;
}

static MidArgs mid_args(fb_model *m, int g, int stage);
static int full_import_state(fb_model *m, cf *spec3);
static int full_export_state(fb_model *m, cf *dst);
static int launch_rowh2(fb_ctx *c, const RowArgs &a);

extern "C" int fb_model_set_vort(fb_model *m, const float *d_vort)
{
    if (!m || !d_vort) return fail(FB_EINVAL, "fb_model_set_vort: NULL");
    fb_ctx *c = m->c;
    NEED_SINGLE(c);
    m->warmed = false;                                      // the next fb_model_step starts with an eager (priming) step
    cf *dst = m->gb[0].ZB;                                  // 3-pass row layout in ZB (stage scratch), then into ZA's layout
    HIPCHK(hipMemsetAsync(dst, 0, priv_elems(c) * sizeof(cf), c->stream));
    m->primed = 0;
    int rc = r2c_private(c, d_vort, dst);                   // main.cpp:256
    if (rc) return rc;
    if (m->full) return full_import_state(m, dst);
    return state_convert(c, c->grp[0], dst, m->gb[0].ZA, true);
}

extern "C" int fb_model_set_source(fb_model *m, const float *d_src)
{
    if (!m) return fail(FB_EINVAL, "model NULL");
    fb_ctx *c = m->c;
    const size_t n = (size_t)c->XL * c->ny * sizeof(float);       // the caller's local rows
    // NULL: no source.  The row kernels test m->src alone, so the row flags (XL ints) are kept for the next source.
    if (!d_src) { if (m->src) { HIPCHK(hipStreamSynchronize(c->stream)); hipFree(m->src); m->src = nullptr; } return FB_OK; }
    if (!m->src_nz && hipMalloc((void **)&m->src_nz, (size_t)c->XL * sizeof(int)) != hipSuccess) { m->src_nz = nullptr; return fail(FB_ENOMEM, "source allocation failed"); }
    // the flags exist before the source is published: a model that goes on stepping after FB_ENOMEM never pairs a source with NULL flags
    if (!m->src && hipMalloc((void **)&m->src, n) != hipSuccess) { m->src = nullptr; return fail(FB_ENOMEM, "source allocation failed"); }
    hipLaunchKernelGGL(k_src_row_flags, dim3(c->XL), dim3(256), 0, c->stream, d_src, m->src_nz, c->ny);
    HIPCHK(hipGetLastError());
    if (c->use_rowq) {                                            // k_rowq reads vort_src in its own physical-space order
        hipLaunchKernelGGL(k_rowq_permute_src, dim3(grid_for(c, (size_t)c->XL * c->ny / 2)), dim3(256), 0, c->stream, d_src, m->src, c->XL);
        HIPCHK(hipGetLastError());
        return FB_OK;
    }
    if (c->rowh_v) {                                              // k_rowh reads vort_src in its own physical-space order
        const size_t pairs = (size_t)c->XL * c->ny / 2;
        if (c->rowh_v == 1) hipLaunchKernelGGL((k_rowh_permute_src<1>), dim3(grid_for(c, pairs)), dim3(256), 0, c->stream, d_src, m->src, c->XL);
        else hipLaunchKernelGGL((k_rowh_permute_src<2>), dim3(grid_for(c, pairs)), dim3(256), 0, c->stream, d_src, m->src, c->XL);
        HIPCHK(hipGetLastError());
        return FB_OK;
    }
    HIPCHK(hipMemcpyAsync(m->src, d_src, n, hipMemcpyDeviceToDevice, c->stream));
    return FB_OK;
}

static MidArgs mid_args(fb_model *m, int g, int stage)
{
    fb_ctx *c = m->c;
    const ColGroup &G = c->grp[g];
    const GroupBufs &B = m->gb[g];
    MidArgs a;
    a.Tin = B.t_recv; a.Zbase = B.ZA; a.Zcur = B.ZB; a.Acc = B.ACC; a.Zout = B.ZA; a.W4 = B.w4_send;
    a.ct0 = 0; a.nct = stage < 0 ? G.ncols / 16 : G.nct_active;     // priming covers every column once
    a.fstride = w4_fstride(c, G); a.rm = rowmap_w4(c, G); a.P = G.ncols; a.N1 = c->N1; a.N2 = c->N2; a.ky0 = G.ky0; a.stage = stage;
    a.nu = m->nu; a.dt = m->dt; a.coef = make_coef(c); a.tw_n = c->d_tw_n2; a.tw_big = c->d_tw_big;
    return a;
}


static int launch_col_full(fb_model *m, int stage);

// full-path model: vort_c arrives in the 3-pass layout in `spec3`: move it into k_col_full's layout (every column, the
// frozen ky = ny/2 one included) and run the PRIME launch, which leaves the four derivative fields of every column in W4
static int full_import_state(fb_model *m, cf *spec3)
{
    fb_ctx *c = m->c;
    const int ntiles = (c->ny / 2) / 8;
    hipLaunchKernelGGL((k_full_relayout<true>), dim3(c->max_wg), dim3(256), 0, c->stream, (const cf *)spec3, m->gb[0].ZA, c->P, c->N1, c->N2, ntiles, m->nsub, c->hy);
    HIPCHK(hipGetLastError());
    int rc = launch_col_full(m, 4);
    if (rc) return rc;
    m->primed = 2;
    return FB_OK;
}

// full-path model: vort_c -> 3-pass layout in `dst` (nx*P complex; pad columns zeroed)
static int full_export_state(fb_model *m, cf *dst)
{
    fb_ctx *c = m->c;
    HIPCHK(hipMemsetAsync(dst, 0, priv_elems(c) * sizeof(cf), c->stream));
    const int ntiles = (c->ny / 2) / 8;
    hipLaunchKernelGGL((k_full_relayout<false>), dim3(c->max_wg), dim3(256), 0, c->stream, (const cf *)m->gb[0].ZA, dst, c->P, c->N1, c->N2, ntiles, m->nsub, c->hy);
    HIPCHK(hipGetLastError());
    return FB_OK;
}

// stage 0..3: forward x transform of the tendency + RK update + derivatives; stage 4: PRIME (derivatives of vort_c only)
static int launch_col_full(fb_model *m, int stage)
{
    fb_ctx *c = m->c;
    const GroupBufs &B = m->gb[0];
    FullArgs a;
    a.Tin = B.t_recv; a.Zbase = B.ZA; a.Zcur = B.ZB; a.Acc = B.ACC; a.Zout = B.ZA; a.W4 = B.w4_send;
    a.fstride = (long)priv_elems(c); a.P = c->P; a.ntiles = (c->ny / 2) / 8; a.stage = stage; a.nu = m->nu; a.dt = m->dt;
    a.nsub = m->nsub; a.sub_rows = 4096;
    a.wscale = m->prescale ? 1.0f / (float)((size_t)c->nx * c->ny) : 1.0f;
    a.ntiles_run = c->grp[0].nct_active * 2 < a.ntiles ? c->grp[0].nct_active * 2 : a.ntiles;   // 16-column tiles -> 8-column tiles
    if (getenv("FB_FULL_NOSKIP")) a.ntiles_run = a.ntiles;
    if (stage == 4) a.ntiles_run = a.ntiles + 1;
    a.coef = make_coef(c); a.tw256 = c->d_tw_256; a.tw4096 = m->nsub == 1 ? c->d_tw_big : c->d_tw_4096;
    const dim3 g(m->nsub * (stage == 4 ? a.ntiles + 1 : a.ntiles)), b(CF_THREADS);
#define FB_LAUNCH_FULL(ST) do { if (m->nsub == 1) hipLaunchKernelGGL((k_col_full<ST, 1>), g, b, CF_LDS_BYTES, c->stream, a); \
                                else hipLaunchKernelGGL((k_col_full<ST, 2>), g, b, CF_LDS_BYTES, c->stream, a); } while (0)
    switch (stage) {
    case 0: FB_LAUNCH_FULL(0); break;
    case 1: FB_LAUNCH_FULL(1); break;
    case 2: FB_LAUNCH_FULL(2); break;
    case 3: FB_LAUNCH_FULL(3); break;
    default: FB_LAUNCH_FULL(4); break;
    }
#undef FB_LAUNCH_FULL
    HIPCHK(hipGetLastError());
    return FB_OK;
}

// ---- the local passes of one RK stage, shared by the fused single-GPU flow and the multi-GPU driver ----------------
// priming: derivatives of vort_c for every column of every group; the frozen tiles get their backward strided
// sub-pass here, once -- the per-stage passes only touch the active tiles of group 0
static int model_prime(fb_model *m)
{
    fb_ctx *c = m->c;
    int rc;
    for (int g = 0; g < c->ngroups; ++g) {
        const ColGroup &G = c->grp[g];
        if ((rc = launch_col_mid(c, mid_args(m, g, -1)))) return rc;
        if ((rc = launch_col_strided<+1>(c, G, m->gb[g].w4_send, 4, w4_fstride(c, G), rowmap_w4(c, G), G.nct_active, G.ncols / 16 - G.nct_active))) return rc;
    }
    m->primed = 1;
    return FB_OK;
}
// backward strided sub-pass on the active tiles, fields [f0, f1): the multi-GPU step pipelines it field by field against
// the exchange; the fused flow runs it once right after priming (its stage loop starts with the row pass)
static int model_col_bwd_active(fb_model *m, int f0 = 0, int f1 = 4, int g = 0)
{
    fb_ctx *c = m->c;
    const ColGroup &G = c->grp[g];
    return launch_col_strided<+1>(c, G, m->gb[g].w4_send + (size_t)f0 * w4_fstride(c, G), f1 - f0, w4_fstride(c, G), rowmap_w4(c, G), 0, G.nct_active);
}
static RowArgs fused_row_args(fb_model *m, int x0, int nrows)
{
    fb_ctx *c = m->c;
    RowArgs a = row_args_base(c);
    if (c->world == 1) { a.M = view_single(c, m->gb[0].w4_recv, (long)priv_elems(c)); a.T = view_single(c, m->gb[0].t_send, 0); }
    else {
        const cf *w4[3] = {m->gb[0].w4_recv, m->gb[1].w4_recv, m->gb[2].w4_recv}, *ts[3] = {m->gb[0].t_send, m->gb[1].t_send, m->gb[2].t_send};
        a.M = view_slab(c, w4, 4); a.T = view_slab(c, ts, 1); a.t_frozen = 0;
    }
    a.src = m->src; a.src_nz = m->src_nz; a.scale = 1.0f / (float)((size_t)c->nx * c->ny); a.x0 = x0; a.nx = nrows; a.prescaled = m->prescale ? 1 : 0;
    return a;
}
// forward x pass of the tendency + RK stage update + derivatives of the new stage state (three-kernel path)
static int model_col_fwd(fb_model *m, int stage, int g = 0)
{
    fb_ctx *c = m->c;
    int rc;
    if ((rc = launch_col_strided<-1>(c, c->grp[g], m->gb[g].t_recv, 1, 0, rowmap_natural(), 0, c->grp[g].nct_active))) return rc;
    m->primed = 1;
    return launch_col_mid(c, mid_args(m, g, stage));
}

// optional per-launch HIP-event profiler (bench.py's roofline leg)
struct StepProf {
    std::vector<hipEvent_t> ev0[4], ev1[4];
    hipStream_t stream;
    int begin(int cls) { hipEvent_t a, b; if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return 1;
                         ev0[cls].push_back(a); ev1[cls].push_back(b); return hipEventRecord(a, stream) != hipSuccess; }
    int end(int cls) { return hipEventRecord(ev1[cls].back(), stream) != hipSuccess; }
};
#define PROF_BEGIN(cls) do { if (prof && prof->begin(cls)) return fail(FB_EHIP, "event record failed"); } while (0)
#define PROF_END(cls) do { if (prof && prof->end(cls)) return fail(FB_EHIP, "event record failed"); } while (0)

static int model_step_impl(fb_model *m, int nsteps, StepProf *prof)
{
    if (!m || nsteps < 0) return fail(FB_EINVAL, "fb_model_step: bad argument");
    fb_ctx *c = m->c;
    if (c->world != 1 || m->phase_flow) return fail(FB_EINVAL, "fb_model_step on a slab model: drive it with fb_slab_step");
    int rc;
    if (nsteps == 0) return FB_OK;
    if (m->full && !m->primed) return fail(FB_EINVAL, "fb_model_step: set the state first");
    if (!m->primed && (rc = model_prime(m))) return rc;
    if (m->primed == 1 && !m->full) { if ((rc = model_col_bwd_active(m))) return rc; m->primed = 2; }
    const ColGroup &G = c->grp[0];
    GroupBufs &B = m->gb[0];
    for (int s = 0; s < nsteps; ++s) {
        for (int k = 0; k < 4; ++k) {
            // row pass on the derivative fields left by the previous stage (or the priming pass) ...
            RowArgs a = fused_row_args(m, 0, c->XL);
            PROF_BEGIN(1);
            if (m->full && m->nsub == 2) {                // one workgroup per x2 produces the rows x2 and x2 + 4096 (k_rowh2)
                a.nx = 4096; a.sub_rows = 4096; a.tw_x = c->d_tw_big;
                if ((rc = launch_rowh2(c, a))) return rc;
            } else if ((rc = launch_row<ROW_FUSED>(c, a))) return rc;
            PROF_END(1);
            if (m->full) {                        // two launches per stage: row pass, single-pass x transform
                PROF_BEGIN(3);
                if ((rc = launch_col_full(m, k))) return rc;
                PROF_END(3);
                continue;
            }
            // ... then the x pass in column chunks, each chained forward -> update -> backward so that a
            // chunk's derivative fields are still in the Infinity Cache when the backward sub-pass reads them
            const int nchunk = c->col_chunks;
            // chunks are independent of each other: spread over several streams their kernels overlap, and
            // the drain of one kernel is filled by the next (not while profiling: the events would interleave)
            const int nstr = (prof || m->use_graph) ? 1 : c->col_streams;
            hipStream_t main_stream = c->stream;
            if (nstr > 1) {
                if (!c->ev_fork) HIPCHK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
                for (int i = 0; i < nstr - 1; ++i) {
                    if (!c->aux[i]) HIPCHK(hipStreamCreateWithFlags(&c->aux[i], hipStreamNonBlocking));
                    if (!c->ev_join[i]) HIPCHK(hipEventCreateWithFlags(&c->ev_join[i], hipEventDisableTiming));
                }
                HIPCHK(hipEventRecord(c->ev_fork, main_stream));
                for (int i = 0; i < nstr - 1; ++i) HIPCHK(hipStreamWaitEvent(c->aux[i], c->ev_fork, 0));
            }
            struct StreamGuard { fb_ctx *c; hipStream_t s; ~StreamGuard() { c->stream = s; } } guard{c, main_stream};
            for (int h = 0; h < nchunk; ++h) {
                const int ct0 = (int)((long)G.nct_active * h / nchunk), ct1 = (int)((long)G.nct_active * (h + 1) / nchunk);
                if (ct1 == ct0) continue;
                c->stream = (nstr > 1 && h % nstr) ? c->aux[h % nstr - 1] : main_stream;
                PROF_BEGIN(2);
                if ((rc = launch_col_strided<-1>(c, G, B.t_recv, 1, 0, rowmap_natural(), ct0, ct1 - ct0))) return rc;
                PROF_END(2);
                MidArgs ma = mid_args(m, 0, k); ma.ct0 = ct0; ma.nct = ct1 - ct0;
                PROF_BEGIN(3);
                if ((rc = launch_col_mid(c, ma))) return rc;
                PROF_END(3);
                PROF_BEGIN(0);
                if ((rc = launch_col_strided<+1>(c, G, B.w4_send, 4, (long)priv_elems(c), rowmap_natural(), ct0, ct1 - ct0))) return rc;
                PROF_END(0);
            }
            c->stream = main_stream;
            for (int i = 0; i < nstr - 1; ++i) {
                HIPCHK(hipEventRecord(c->ev_join[i], c->aux[i]));
                HIPCHK(hipStreamWaitEvent(main_stream, c->ev_join[i], 0));
            }
        }
    }
    return FB_OK;
}

static void model_drop_graph(fb_model *m)
{
    if (m->graph_exec) { hipGraphExecDestroy(m->graph_exec); m->graph_exec = nullptr; }
}

extern "C" int fb_model_use_graph(fb_model *m, int enable)
{
    if (!m) return fail(FB_EINVAL, "model NULL");
    m->use_graph = enable != 0;
    if (!m->use_graph) model_drop_graph(m);
    return FB_OK;
}

extern "C" int fb_model_step(fb_model *m, int nsteps)
{
    if (!m || nsteps < 0) return fail(FB_EINVAL, "fb_model_step: bad argument");
    fb_ctx *c = m->c;
    // graph replay needs a capturable (non-null) stream, a primed pipeline and at least one eager step behind
    // us (kernel attributes are set on first launch); anything else runs eagerly
    if (!m->use_graph || c->world != 1 || c->stream == nullptr || nsteps < 2) { if (nsteps > 0) m->warmed = true; return model_step_impl(m, nsteps, nullptr); }
    int rc;
    if (!m->warmed) { if ((rc = model_step_impl(m, 1, nullptr))) return rc; m->warmed = true; --nsteps; }
    if (m->graph_exec && (m->graph_src != m->src || m->graph_stream != c->stream)) model_drop_graph(m);
    if (!m->graph_exec) {
        hipGraph_t g = nullptr;
        HIPCHK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
        rc = model_step_impl(m, 1, nullptr);
        hipError_t e = hipStreamEndCapture(c->stream, &g);
        if (rc) { if (g) hipGraphDestroy(g); return rc; }
        HIPCHK(e);
        e = hipGraphInstantiate(&m->graph_exec, g, nullptr, nullptr, 0);
        hipGraphDestroy(g);
        HIPCHK(e);
        m->graph_src = m->src; m->graph_stream = c->stream;
        // the captured step has NOT executed: replay it below like the others
    }
    for (int s = 0; s < nsteps; ++s) HIPCHK(hipGraphLaunch(m->graph_exec, c->stream));
    return FB_OK;
}

extern "C" int fb_model_time_steps(fb_model *m, int nsteps, float *total_ms)
{
    if (!m || !total_ms) return fail(FB_EINVAL, "fb_model_time_steps: NULL");
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipEventRecord(e0, m->c->stream));
    int rc = fb_model_step(m, nsteps);
    HIPCHK(hipEventRecord(e1, m->c->stream));
    HIPCHK(hipEventSynchronize(e1));
    HIPCHK(hipEventElapsedTime(total_ms, e0, e1));
    hipEventDestroy(e0); hipEventDestroy(e1);
    return rc;
}

extern "C" int fb_model_profile_steps(fb_model *m, int nsteps, float *ms_sum, int *launches)
{
    if (!m || !ms_sum || !launches) return fail(FB_EINVAL, "fb_model_profile_steps: NULL");
    StepProf prof; prof.stream = m->c->stream;
    if (!m->primed) { int rc0 = model_step_impl(m, 0, nullptr); (void)rc0; }
    int rc = model_step_impl(m, nsteps, &prof);
    HIPCHK(hipStreamSynchronize(m->c->stream));
    for (int cls = 0; cls < 4; ++cls) {
        ms_sum[cls] = 0.f; launches[cls] = (int)prof.ev0[cls].size();
        for (size_t i = 0; i < prof.ev0[cls].size(); ++i) {
            float ms = 0.f;
            if (i < prof.ev1[cls].size() && hipEventElapsedTime(&ms, prof.ev0[cls][i], prof.ev1[cls][i]) == hipSuccess) ms_sum[cls] += ms;
            hipEventDestroy(prof.ev0[cls][i]);
            if (i < prof.ev1[cls].size()) hipEventDestroy(prof.ev1[cls][i]);
        }
    }
    return rc;
}

extern "C" int fb_model_get_spectrum(fb_model *m, float *d_spec)
{
    if (!m || !d_spec) return fail(FB_EINVAL, "fb_model_get_spectrum: NULL");
    NEED_SINGLE(m->c);
    int rc;
    if ((rc = ensure_scratch(m->c))) return rc;
    if (m->full) { if ((rc = full_export_state(m, m->c->d_scratch))) return rc; }
    else if ((rc = state_convert(m->c, m->c->grp[0], m->gb[0].ZA, m->c->d_scratch, false))) return rc;
    return relayout(m->c, m->c->d_scratch, (cf *)d_spec, false);
}

extern "C" int fb_model_set_spectrum(fb_model *m, const float *d_spec)
{
    if (!m || !d_spec) return fail(FB_EINVAL, "fb_model_set_spectrum: NULL");
    NEED_SINGLE(m->c);
    m->primed = 0;
    m->warmed = false;
    int rc = relayout(m->c, (const cf *)d_spec, m->gb[0].ZB, true);
    if (rc) return rc;
    if (m->full) return full_import_state(m, m->gb[0].ZB);
    return state_convert(m->c, m->c->grp[0], m->gb[0].ZB, m->gb[0].ZA, true);
}

extern "C" int fb_model_get_vort(fb_model *m, float *d_vort)
{
    if (!m || !d_vort) return fail(FB_EINVAL, "fb_model_get_vort: NULL");
    fb_ctx *c = m->c;
    NEED_SINGLE(c);
    int rc;
    if ((rc = ensure_scratch(c))) return rc;
    // copy of vort_c (main.cpp:273), c2r, normalise (main.cpp:275)
    if (m->full) { if ((rc = full_export_state(m, c->d_scratch))) return rc; }
    else if ((rc = state_convert(c, c->grp[0], m->gb[0].ZA, c->d_scratch, false))) return rc;
    return c2r_private(c, c->d_scratch, d_vort, 1.0f / (float)((size_t)c->nx * c->ny));
}

extern "C" int fb_model_get_diag(fb_model *m, float *d_psi, float *d_u, float *d_v)
{
    if (!m) return fail(FB_EINVAL, "model NULL");
    fb_ctx *c = m->c;
    NEED_SINGLE(c);
    const size_t n = (size_t)c->nx * c->hy * sizeof(cf);
    for (auto &p : m->nat)
        if (!p && hipMalloc((void **)&p, n) != hipSuccess) return fail(FB_ENOMEM, "record-path allocation failed");
    int rc;
    float *vc = (float *)m->nat[0], *psi = (float *)m->nat[1], *tmp = (float *)m->nat[2];
    if ((rc = fb_model_get_spectrum(m, vc))) return rc;
    if ((rc = fb_invert_laplacian(c, vc, psi))) return rc;                       // main.cpp:179
    if (d_psi && (rc = fb_c2r(c, psi, d_psi, 1))) return rc;                     // :185-188
    if (d_u) {
        if ((rc = fb_grady(c, psi, tmp)) || (rc = fb_c2r(c, tmp, d_u, 1)) || (rc = fb_negate(c, d_u))) return rc;   // :198-201
    }
    if (d_v) {
        if ((rc = fb_gradx(c, psi, tmp)) || (rc = fb_c2r(c, tmp, d_v, 1))) return rc;                                 // :212-214
    }
    return FB_OK;
}

#include "fb_slab_driver.h"

// field I/O (fb_write_field / fb_read_field, writeField / readField): fb_fieldio.cpp
