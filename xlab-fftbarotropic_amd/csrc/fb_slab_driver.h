// fb_slab_driver.h -- the multi-GPU RK4 step, driven from C (included by fftbaro.hip; C ABI: include/fftbaro.h, fb_slab_*).
//
// One process per GPU: physical / mixed fields are split by x rows (XL = nx/world per rank),
// spectral fields by ky columns -- the ACTIVE columns (ky < world*KA, at least one mode inside the dealiasing circle)
// evenly over the ranks, the FROZEN ones beyond them likewise (ColGroup, fftbaro.hip).  Per RK stage two all-to-all
// transposes: the four derivative fields (columns -> rows) and the tendency (rows -> columns), active columns only; the
// frozen columns' derivative fields cross the links once, at priming, and their tendency is never sent (it is masked,
// SURVEY note N1).  No reference counterpart: the reference is single-process (main.cpp:259-323 is what one step computes).
//
// Schedule of one stage, compute stream S and communication stream C (events in between).  Small slabs (one active column group):
//   S: backward strided x sub-pass of field group 0 | group 1 | ...            C: all-to-all(group 0) | all-to-all(group 1) ...
//   S: row pass of row chunk 0 | chunk 1 | ...                                  C: all-to-all(tendency chunk 0) | chunk 1 ...
//   S: forward x pass + RK update + derivatives (needs every row)
// so the links are busy from the end of the first sub-pass to the arrival of the last tendency chunk, and the forward pass, the
// first sub-pass and the first row chunk are exposed.  Large slabs (BASELINE configs 4 and 5): the active columns of a rank are
// cut into TWO column groups A, B with exchange buffers of their own (fb_ctx::grp), and the stage runs
//   S: row chunk 0 | 1 | ...     | x pass(A): forward, update, derivatives, backward | x pass(B)            | row chunk 0 ...
//   C:     a2a T_A(0), T_B(0) | ...  T_A(last) | T_B(last)                 | a2a W4_A                | a2a W4_B |
// x pass(A) starts when A's tendency is in (B's is still on the links) and A's four derivative fields leave while B is
// computed: between the last tendency chunk and the first derivative message the links never wait for a kernel, and only the
// first row chunk is exposed.  The granularity follows the local work a piece hides (fb_slab_plan, slab_active_groups): a
// piece that hides less than a collective's latency is not cut off.
#pragma once
#include "fb_transport.h"

struct fb_slab {
    fb_ctx *c;
    fb_model *m;
    fb_transport tp;
    bool connected, owns_streams;
    hipStream_t comp, comm;
    hipEvent_t ev_f[4], ev_r[8], ev_w4, ev_t, ev_tg[2], ev_rows_done, ev_fwd_done, ev_misc[2];
    int nfg, nch;               // field groups of the derivative exchange (1, 2 or 4), row chunks of the tendency exchange (1..8)
    int ncg;                    // active column groups (fb_ctx::nact): 2 = the stage is pipelined by column groups, and nfg == 1
    int step_ops;               // exchange operations issued per RK stage (diagnostics)
};

// ---- schedule (host logic, no GPU needed): how finely one stage's two transposes are pipelined ----
#define FB_OP_COL_BWD   1       /* arg = field group */
#define FB_OP_XCHG_W4   2       /* arg = field group */
#define FB_OP_ROW       3       /* arg = row chunk   */
#define FB_OP_XCHG_T    4       /* arg = row chunk   */
#define FB_OP_COL_FWD   5       /* arg = column group (0 unless the stage is pipelined by column groups) */
#define FB_OP_COL_ALL_BWD 6     /* backward x sub-pass of all four fields of column group arg */
static void slab_plan(int nx, int ny, int world, int *nfg, int *nch, int *ncg = nullptr)
{
    const int dxw = (int)ceil(((double)(float)nx) / 3.0), dyw = (int)ceil(((double)(float)ny) / 3.0);
    int jmax, KA, KF;
    slab_split(ny, (double)(float)((double)dxw * dxw + (double)dyw * dyw), world, jmax, KA, KF);
    const long XL = nx / world;
    // Pipelining a transpose against the pass that feeds it hides that pass's time, minus one more collective's latency per
    // extra piece (tens of microseconds for a grouped RCCL send/recv).  So the derivative exchange is cut by fields only where one
    // field's backward sub-pass (2 * nx * KA * 8 bytes at ~5 TB/s) is worth an operation, and the tendency exchange by row chunks
    // only where half the row pass (5 * XL * (ny/2+1) * 8 bytes at ~4 TB/s) is.
    const double bwd_us = 2.0 * nx * KA * 8.0 / 5e6, row_us = 5.0 * XL * (ny / 2 + 1) * 8.0 / 4e6;
    int fg = world == 1 ? 1 : (bwd_us >= 20.0 ? 4 : (bwd_us >= 10.0 ? 2 : 1));
    int ch = world == 1 ? 1 : (row_us >= 100.0 ? 2 : 1);
    if (const char *e = getenv("FB_SLAB_FIELD_GROUPS")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4) fg = v; }
    if (const char *e = getenv("FB_SLAB_ROW_CHUNKS")) { const int v = atoi(e); if (v >= 1 && v <= 8) ch = v; }
    while (ch > 1 && ((XL / ch) & 1 || XL % ch)) ch >>= 1;          // chunks are whole row pairs
    const int cg = slab_active_groups(nx, world, KA);
    if (cg > 1) fg = 1;                                             // pipelined by column groups instead: a group's four fields leave together
    *nfg = fg; *nch = ch;
    if (ncg) *ncg = cg;
}

extern "C" int fb_slab_plan(int nx, int ny, int world, int *field_groups, int *row_chunks, int *ops, int cap)
{
    if (!fb_size_supported(nx, ny) || world < 1 || !is_pow2(world) || nx / world < 2) { fail(FB_EINVAL, "fb_slab_plan: bad geometry"); return 0; }
    int nfg, nch, ncg;
    slab_plan(nx, ny, world, &nfg, &nch, &ncg);
    if (field_groups) *field_groups = nfg;
    if (row_chunks) *row_chunks = nch;
    int n = 0;
    auto put = [&](int op, int arg) { if (ops && n < cap) ops[n] = op * 16 + arg; ++n; };
    if (ncg > 1) {                                         // pipelined by column groups (arguments of kinds 2 and 5/6: the column group)
        for (int h = 0; h < nch; ++h) { put(FB_OP_ROW, h); put(FB_OP_XCHG_T, h); }
        for (int g = 0; g < ncg; ++g) { put(FB_OP_COL_FWD, g); put(FB_OP_COL_ALL_BWD, g); put(FB_OP_XCHG_W4, g); }
        return n;
    }
    for (int g = 0; g < nfg; ++g) { put(FB_OP_COL_BWD, g); if (world > 1) put(FB_OP_XCHG_W4, g); }
    for (int h = 0; h < nch; ++h) { put(FB_OP_ROW, h); if (world > 1) put(FB_OP_XCHG_T, h); }
    put(FB_OP_COL_FWD, 0);
    return n;                                              // number of operations of one RK stage (>= 0, not a status)
}

extern "C" int fb_slab_geometry(int nx, int ny, int world, int *rows_local, int *cols_active, int *cols_frozen)
{
    if (!fb_size_supported(nx, ny) || world < 1 || !is_pow2(world) || nx / world < 2) return fail(FB_EINVAL, "fb_slab_geometry: bad geometry");
    const int dxw = (int)ceil(((double)(float)nx) / 3.0), dyw = (int)ceil(((double)(float)ny) / 3.0);
    int jmax, KA, KF;
    slab_split(ny, (double)(float)((double)dxw * dxw + (double)dyw * dyw), world, jmax, KA, KF);
    if (rows_local) *rows_local = nx / world;
    if (cols_active) *cols_active = KA;
    if (cols_frozen) *cols_frozen = KF;
    return FB_OK;
}

// host logic, no GPU needed: how a rank's KA active columns are cut into column groups (1 or 2; whole 16-column tiles, the first
// group takes the odd one).  Rank r's active slab is the global columns [r*KA, (r+1)*KA): group 0 its first cols2[0], group 1 the rest.
extern "C" int fb_slab_col_groups(int nx, int ny, int world, int *ngroups, int *cols2)
{
    if (!fb_size_supported(nx, ny) || world < 1 || !is_pow2(world) || nx / world < 2) return fail(FB_EINVAL, "fb_slab_col_groups: bad geometry");
    const int dxw = (int)ceil(((double)(float)nx) / 3.0), dyw = (int)ceil(((double)(float)ny) / 3.0);
    int jmax, KA, KF;
    slab_split(ny, (double)(float)((double)dxw * dxw + (double)dyw * dyw), world, jmax, KA, KF);
    const int na = world == 1 ? 1 : slab_active_groups(nx, world, KA), tiles = KA / 16;
    if (ngroups) *ngroups = na;
    if (cols2) for (int g = 0; g < 2; ++g) cols2[g] = g < na ? 16 * (tiles / na + (g < tiles % na ? 1 : 0)) : 0;
    return FB_OK;
}

// ---- creation ----
extern "C" int fb_slab_destroy(fb_slab *s)
{
    if (!s) return FB_OK;
    if (s->comp) hipStreamSynchronize(s->comp);
    if (s->comm && s->comm != s->comp) hipStreamSynchronize(s->comm);
    if (s->connected && s->tp.destroy) s->tp.destroy(s->tp.self);
    if (s->m) fb_model_destroy(s->m);
    if (s->c) fb_destroy(s->c);
    hipEvent_t *evs[] = {&s->ev_w4, &s->ev_t, &s->ev_tg[0], &s->ev_tg[1], &s->ev_rows_done, &s->ev_fwd_done, &s->ev_misc[0], &s->ev_misc[1]};
    for (hipEvent_t *e : evs) if (*e) hipEventDestroy(*e);
    for (auto &e : s->ev_f) if (e) hipEventDestroy(e);
    for (auto &e : s->ev_r) if (e) hipEventDestroy(e);
    if (s->owns_streams) { if (s->comm && s->comm != s->comp) hipStreamDestroy(s->comm); if (s->comp) hipStreamDestroy(s->comp); }
    delete s;
    return FB_OK;
}

extern "C" int fb_slab_create(fb_slab **out, int nx, int ny, float lx, float ly, float nu, float dt, int rank, int world)
{
    if (!out) return fail(FB_EINVAL, "fb_slab_create: out is NULL");
    *out = nullptr;
    fb_slab *s = new fb_slab();
    memset(s, 0, sizeof(*s));
    int rc = fb_create_slab(&s->c, nx, ny, lx, ly, rank, world);
    if (rc) { delete s; return rc; }
    auto bail = [&](int code) { fb_slab_destroy(s); return code; };
    if (hipStreamCreateWithFlags(&s->comp, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&s->comm, hipStreamNonBlocking) != hipSuccess)
        return bail(fail(FB_EHIP, "fb_slab_create: cannot create streams"));
    s->owns_streams = true;
    s->c->stream = s->comp;
    hipEvent_t *evs[] = {&s->ev_w4, &s->ev_t, &s->ev_tg[0], &s->ev_tg[1], &s->ev_rows_done, &s->ev_fwd_done, &s->ev_misc[0], &s->ev_misc[1], &s->ev_f[0], &s->ev_f[1], &s->ev_f[2],
                         &s->ev_f[3], &s->ev_r[0], &s->ev_r[1], &s->ev_r[2], &s->ev_r[3], &s->ev_r[4], &s->ev_r[5], &s->ev_r[6], &s->ev_r[7]};
    for (hipEvent_t *e : evs)
        if (hipEventCreateWithFlags(e, hipEventDisableTiming) != hipSuccess) return bail(fail(FB_EHIP, "fb_slab_create: cannot create events"));
    if ((rc = model_create_impl(&s->m, s->c, nu, dt, true))) return bail(rc);
    slab_plan(nx, ny, world, &s->nfg, &s->nch);
    s->ncg = s->c->nact;
    if (s->ncg > 1) s->nfg = 1;
    // Nothing to overlap when a stage has one field group and one row chunk (small slabs): the exchanges then go on the compute
    // stream, in order, and the stage has no cross-stream hand-overs (each costs the GPU ~10-15 us of idling; tools/slab_local_time.py:
    // rank-local 4096^2 step on 8 ranks 0.47 -> 0.3x ms).  FB_SLAB_TWO_STREAMS=1 keeps the separate communication stream.
    if (s->nfg == 1 && s->nch == 1 && s->ncg == 1 && !getenv("FB_SLAB_TWO_STREAMS")) {
        hipStreamDestroy(s->comm);
        s->comm = s->comp;
    }
    s->connected = world == 1;                             // nothing to connect on one rank
    *out = s;
    return FB_OK;
}

static int slab_connected(fb_slab *s, int rc)
{
    if (rc) return rc;
    s->connected = true;
    return FB_OK;
}
#define SLAB_CONNECT_GUARD(s) do { if (!(s)) return fail(FB_EINVAL, "slab NULL"); if ((s)->connected && (s)->c->world > 1) return fail(FB_EINVAL, "slab already connected"); } while (0)
extern "C" int fb_slab_connect_rccl(fb_slab *s, const char *unique_id)
{
    SLAB_CONNECT_GUARD(s);
    if (!unique_id) return fail(FB_EINVAL, "fb_slab_connect_rccl: NULL id");
    return slab_connected(s, fb_transport_rccl(&s->tp, unique_id, s->c->rank, s->c->world));
}
extern "C" int fb_slab_connect_local(fb_slab *s, void *hub)
{
    SLAB_CONNECT_GUARD(s);
    return slab_connected(s, fb_transport_local(&s->tp, hub, s->c->rank, s->c->world));
}
extern "C" int fb_slab_connect_callback(fb_slab *s, fb_alltoall_fn fn, void *user)
{
    SLAB_CONNECT_GUARD(s);
    return slab_connected(s, fb_transport_callback(&s->tp, fn, user, s->c->rank, s->c->world));
}

extern "C" int fb_slab_info(fb_slab *s, int *rows_local, int *cols_active, int *cols_frozen, int *ky0_active, int *ky0_frozen,
                            int *field_groups, int *row_chunks)
{
    if (!s) return fail(FB_EINVAL, "slab NULL");
    const fb_ctx *c = s->c;
    if (rows_local) *rows_local = c->XL;
    if (cols_active) *cols_active = c->world > 1 ? c->KA : c->grp[0].ncols;        // every active column group together
    if (cols_frozen) *cols_frozen = c->ngroups > c->nact ? c->grp[c->nact].ncols : 0;
    if (ky0_active) *ky0_active = c->grp[0].ky0;
    if (ky0_frozen) *ky0_frozen = c->ngroups > c->nact ? c->grp[c->nact].ky0 : 0;
    if (field_groups) *field_groups = s->nfg;
    if (row_chunks) *row_chunks = s->nch;
    return FB_OK;
}

// What is connected: the transport's name, what ITS communicator reports (RCCL: ncclCommCount, ncclCommUserRank, ncclCommCuDevice;
// -1 for transports without one) and the HIP device this rank computes on.
extern "C" int fb_slab_transport_info(fb_slab *s, char *name, size_t cap, int *comm_ranks, int *comm_rank, int *comm_device, int *hip_device)
{
    if (!s) return fail(FB_EINVAL, "slab NULL");
    // a transport that is connected is named, also on one rank (the one-GPU test of the RCCL call path connects one); "none" = one rank, nothing connected
    const char *nm = (s->tp.alltoall && s->tp.name) ? s->tp.name : (s->c->world == 1 ? "none" : "unconnected");
    if (name && cap) { strncpy(name, nm, cap - 1); name[cap - 1] = 0; }
    if (comm_ranks) *comm_ranks = -1;
    if (comm_rank) *comm_rank = -1;
    if (comm_device) *comm_device = -1;
    if (s->tp.alltoall && s->tp.info) s->tp.info(s->tp.self, comm_ranks, comm_rank, comm_device);
    if (hip_device) { int d = -1; if (hipGetDevice(&d) != hipSuccess) d = -1; *hip_device = d; }
    return FB_OK;
}

// Exchanges a known pattern through the connected transport (world*count floats each way) and returns the number of wrong
// words: a start-up check of the links, and the one-GPU test of the RCCL call path (world = 1 with FB_RCCL_SELF=1).
__global__ void k_slab_pattern(float *buf, size_t count, int world, int rank, int check, unsigned long long *bad)
{
    const size_t n = (size_t)world * count;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int p = (int)(i / count);
        const size_t j = i - (size_t)p * count;
        // send: block p carries (sender, receiver = p, index); recv: block p must carry (sender = p, receiver = rank, index)
        const float want = check ? (float)(p * 131 + rank * 17) + (float)(j % 1021) : (float)(rank * 131 + p * 17) + (float)(j % 1021);
        if (!check) buf[i] = want; else if (buf[i] != want) atomicAdd(bad, 1ull);
    }
}
extern "C" int fb_slab_transport_selftest(fb_slab *s, size_t count, size_t *wrong_words)
{
    if (!s || !wrong_words || count == 0) return fail(FB_EINVAL, "fb_slab_transport_selftest: bad argument");
    if (!s->tp.alltoall) return fail(FB_EINVAL, "fb_slab_transport_selftest: no transport connected");
    const int W = s->c->world;
    float *snd = nullptr, *rcv = nullptr; unsigned long long *bad = nullptr, hbad = 0;
    int rc = FB_OK;
    if (hipMalloc((void **)&snd, W * count * sizeof(float)) != hipSuccess || hipMalloc((void **)&rcv, W * count * sizeof(float)) != hipSuccess ||
        hipMalloc((void **)&bad, sizeof(*bad)) != hipSuccess) rc = fail(FB_ENOMEM, "selftest allocation failed");
    if (!rc) {
        hipMemsetAsync(bad, 0, sizeof(*bad), s->comm);
        hipMemsetAsync(rcv, 0xff, W * count * sizeof(float), s->comm);
        hipLaunchKernelGGL(k_slab_pattern, dim3(256), dim3(256), 0, s->comm, snd, count, W, s->c->rank, 0, bad);
        rc = s->tp.alltoall(s->tp.self, snd, rcv, count, 0, count, s->comm);
        if (!rc) {
            hipLaunchKernelGGL(k_slab_pattern, dim3(256), dim3(256), 0, s->comm, rcv, count, W, s->c->rank, 1, bad);
            if (hipMemcpyAsync(&hbad, bad, sizeof(hbad), hipMemcpyDeviceToHost, s->comm) != hipSuccess || hipStreamSynchronize(s->comm) != hipSuccess)
                rc = fail(FB_EHIP, "selftest: device error");
        }
    }
    if (snd) hipFree(snd);
    if (rcv) hipFree(rcv);
    if (bad) hipFree(bad);
    *wrong_words = (size_t)hbad;
    return rc;
}

// ---- stream plumbing ----
static int slab_after(hipStream_t waiter, hipStream_t producer, hipEvent_t ev)      // `waiter` continues behind what `producer` has queued so far
{
    if (waiter == producer) return FB_OK;                 // one stream: it does anyway
    HIPCHK(hipEventRecord(ev, producer));
    HIPCHK(hipStreamWaitEvent(waiter, ev, 0));
    return FB_OK;
}
// all-to-all on the communication stream; units are complex elements
static int slab_xchg(fb_slab *s, const cf *send, cf *recv, size_t stride, size_t offset, size_t count)
{
    if (s->c->world == 1 || count == 0) return FB_OK;          // one rank: the passes hand over in place
    ++s->step_ops;
    return s->tp.alltoall(s->tp.self, (const float *)send, (float *)recv, 2 * stride, 2 * offset, 2 * count, s->comm);
}
#define SLAB_READY(s) do { if (!(s)) return fail(FB_EINVAL, "slab NULL"); if (!(s)->connected) return fail(FB_EINVAL, "slab model is not connected to a transport (fb_slab_connect_*)"); } while (0)

// ---- state in / out ----
extern "C" int fb_slab_set_vort_local(fb_slab *s, const float *d_rows)
{
    SLAB_READY(s);
    if (!d_rows) return fail(FB_EINVAL, "fb_slab_set_vort_local: NULL");
    fb_ctx *c = s->c; fb_model *m = s->m;
    int rc;
    // readField + fftwf_execute(p_fwd_vort), main.cpp:143-144,256: y transform of the local rows -> transpose -> x transform of the local columns
    RowArgs a = row_args_base(c);
    a.rin = d_rows;
    const cf *ts[3] = {m->gb[0].t_send, m->gb[1].t_send, m->gb[2].t_send};
    a.T = c->world == 1 ? view_single(c, m->gb[0].t_send, 0) : view_slab(c, ts, 1);
    if ((rc = launch_row<ROW_FWD>(c, a))) return rc;
    if ((rc = slab_after(s->comm, s->comp, s->ev_misc[0]))) return rc;
    for (int g = 0; g < c->ngroups; ++g) {
        const size_t blk = (size_t)c->XL * c->grp[g].ncols;
        if ((rc = slab_xchg(s, m->gb[g].t_send, m->gb[g].t_recv, blk, 0, blk))) return rc;
    }
    if ((rc = slab_after(s->comp, s->comm, s->ev_misc[1]))) return rc;
    for (int g = 0; g < c->ngroups; ++g) {
        const ColGroup &G = c->grp[g];
        if ((rc = launch_col_strided<-1>(c, G, m->gb[g].t_recv, 1, 0)) || (rc = launch_col_block<-1>(c, G, m->gb[g].t_recv, 1, 0))) return rc;
        if ((rc = state_convert(c, G, m->gb[g].t_recv, m->gb[g].ZA, true))) return rc;
    }
    m->primed = 0;
    return FB_OK;
}

extern "C" int fb_slab_set_source_local(fb_slab *s, const float *d_rows)
{
    if (!s) return fail(FB_EINVAL, "slab NULL");
    return fb_model_set_source(s->m, d_rows);               // this rank's [XL][ny] rows of vort_src (main-shallow-water.cpp:304), NULL = zeros
}

// spectral field (a function of vort_c, selected by `what`) -> this rank's rows of the physical field * scale:
// x transform of the local columns -> transpose in the reverse roles -> y transform of the local rows
static int slab_c2r_of_state(fb_slab *s, int what, float scale, float *d_rows)
{
    fb_ctx *c = s->c; fb_model *m = s->m;
    int rc;
    for (int g = 0; g < c->ngroups; ++g) {
        const ColGroup &G = c->grp[g];
        cf *w = m->gb[g].t_recv;
        if ((rc = state_convert(c, G, m->gb[g].ZA, w, false))) return rc;           // copy of vort_c (main.cpp:273)
        const size_t total = grp_elems(c, G);
        const dim3 grid(grid_for(c, total)), blk(256);
        if (what == 1) hipLaunchKernelGGL((k_psi_private<0>), grid, blk, 0, c->stream, make_coef(c), w, G.ncols, c->N1, c->N2, G.ky0);
        else if (what == 2) hipLaunchKernelGGL((k_psi_private<1>), grid, blk, 0, c->stream, make_coef(c), w, G.ncols, c->N1, c->N2, G.ky0);
        else if (what == 3) hipLaunchKernelGGL((k_psi_private<2>), grid, blk, 0, c->stream, make_coef(c), w, G.ncols, c->N1, c->N2, G.ky0);
        HIPCHK(hipGetLastError());
        if ((rc = launch_col_block<+1>(c, G, w, 1, 0)) || (rc = launch_col_strided<+1>(c, G, w, 1, 0))) return rc;   // natural [x][ncols] == [dst][XL][ncols]
    }
    if ((rc = slab_after(s->comm, s->comp, s->ev_misc[0]))) return rc;
    for (int g = 0; g < c->ngroups; ++g) {
        const size_t blk = (size_t)c->XL * c->grp[g].ncols;
        if ((rc = slab_xchg(s, m->gb[g].t_recv, m->gb[g].t_send, blk, 0, blk))) return rc;
    }
    if ((rc = slab_after(s->comp, s->comm, s->ev_misc[1]))) return rc;
    RowArgs a = row_args_base(c);
    const cf *ts[3] = {m->gb[0].t_send, m->gb[1].t_send, m->gb[2].t_send};
    a.M = c->world == 1 ? view_single(c, m->gb[0].t_send, 0) : view_slab(c, ts, 1);
    a.rout = d_rows; a.scale = scale;
    return launch_row<ROW_INV>(c, a);
}

extern "C" int fb_slab_get_vort_local(fb_slab *s, float *d_rows)
{
    SLAB_READY(s);
    if (!d_rows) return fail(FB_EINVAL, "fb_slab_get_vort_local: NULL");
    return slab_c2r_of_state(s, 0, 1.0f / (float)((size_t)s->c->nx * s->c->ny), d_rows);      // record path, main.cpp:273-281
}

// the stage-0 record dumps of main.cpp:181-222 on this rank's rows (any may be NULL): psi, u = -dpsi/dy, v = dpsi/dx
extern "C" int fb_slab_get_diag_local(fb_slab *s, float *d_psi, float *d_u, float *d_v)
{
    SLAB_READY(s);
    const float g = 1.0f / (float)((size_t)s->c->nx * s->c->ny);
    int rc;
    if (d_psi && (rc = slab_c2r_of_state(s, 1, g, d_psi))) return rc;
    if (d_u && (rc = slab_c2r_of_state(s, 2, -g, d_u))) return rc;                  // normalise, then negate (SURVEY note N3): (x * g) * -1 == x * (-g) exactly
    if (d_v && (rc = slab_c2r_of_state(s, 3, g, d_v))) return rc;
    return FB_OK;
}

// ---- the step ----
static int slab_prime(fb_slab *s)
{
    fb_ctx *c = s->c; fb_model *m = s->m;
    int rc;
    if ((rc = model_prime(m))) return rc;                   // derivatives of every column; backward x pass finished on the frozen tiles
    if (c->ngroups > c->nact) {                             // the frozen columns' four fields cross the links once and stay in w4_recv
        const int gf = c->nact;
        const size_t blk = 4 * (size_t)c->XL * c->grp[gf].ncols;
        if ((rc = slab_after(s->comm, s->comp, s->ev_misc[0]))) return rc;
        if ((rc = slab_xchg(s, m->gb[gf].w4_send, m->gb[gf].w4_recv, blk, 0, blk))) return rc;
        if ((rc = slab_after(s->comp, s->comm, s->ev_misc[1]))) return rc;
    }
    return FB_OK;
}

// Pipelined by column groups (ncg == 2).  On entry the derivative fields of every group are complete in w4_recv (m->primed == 2:
// slab_groups_prologue after priming, or the previous stage).
static int slab_stage_groups(fb_slab *s, int stage)
{
    fb_ctx *c = s->c; fb_model *m = s->m;
    int rc;
    const int rows = c->XL / s->nch;
    // row pass (main.cpp:154-237, y part) in row chunks; each chunk's tendency rows leave, group by group, while the next chunk is computed
    for (int h = 0; h < s->nch; ++h) {
        if ((rc = launch_row<ROW_FUSED>(c, fused_row_args(m, h * rows, rows)))) return rc;
        if ((rc = slab_after(s->comm, s->comp, s->ev_r[h]))) return rc;
        for (int g = 0; g < s->ncg; ++g) {
            const size_t nc = c->grp[g].ncols, fld = (size_t)c->XL * nc;
            if ((rc = slab_xchg(s, m->gb[g].t_send, m->gb[g].t_recv, fld, (size_t)h * rows * nc, (size_t)rows * nc))) return rc;
            if (h == s->nch - 1) HIPCHK(hipEventRecord(s->ev_tg[g], s->comm));         // group g's tendency is complete
        }
    }
    // per column group: forward x pass, viscosity, mask, RK stage update, derivatives of the new stage state, backward x pass
    // (main.cpp:148,179-212,237-251,296-312); the group's four fields leave while the next group is computed
    for (int g = 0; g < s->ncg; ++g) {
        const size_t fld = (size_t)c->XL * c->grp[g].ncols;
        HIPCHK(hipStreamWaitEvent(s->comp, s->ev_tg[g], 0));
        if ((rc = model_col_fwd(m, stage, g))) return rc;
        if ((rc = model_col_bwd_active(m, 0, 4, g))) return rc;
        if ((rc = slab_after(s->comm, s->comp, s->ev_f[g]))) return rc;
        if ((rc = slab_xchg(s, m->gb[g].w4_send, m->gb[g].w4_recv, 4 * fld, 0, 4 * fld))) return rc;
    }
    m->primed = 2;
    return slab_after(s->comp, s->comm, s->ev_w4);          // the next row pass (or a record pass) reads w4_recv
}
// after priming: backward x pass on the active tiles of every group and the first exchange of the derivative fields
static int slab_groups_prologue(fb_slab *s)
{
    fb_ctx *c = s->c; fb_model *m = s->m;
    int rc;
    for (int g = 0; g < s->ncg; ++g) {
        const size_t fld = (size_t)c->XL * c->grp[g].ncols;
        if ((rc = model_col_bwd_active(m, 0, 4, g))) return rc;
        if ((rc = slab_after(s->comm, s->comp, s->ev_f[g]))) return rc;
        if ((rc = slab_xchg(s, m->gb[g].w4_send, m->gb[g].w4_recv, 4 * fld, 0, 4 * fld))) return rc;
    }
    m->primed = 2;
    return slab_after(s->comp, s->comm, s->ev_w4);
}

static int slab_stage(fb_slab *s, int stage)
{
    fb_ctx *c = s->c; fb_model *m = s->m;
    if (s->ncg > 1) return slab_stage_groups(s, stage);
    GroupBufs &B = m->gb[0];
    const size_t fld = (size_t)c->XL * c->grp[0].ncols;     // one field's block for one peer
    int rc;
    // derivative fields: finish the backward x pass on the active tiles, field group by field group, and ship each group
    // while the next one is being transformed
    for (int g = 0; g < s->nfg; ++g) {
        const int f0 = 4 * g / s->nfg, f1 = 4 * (g + 1) / s->nfg;
        if ((rc = model_col_bwd_active(m, f0, f1))) return rc;
        if (c->world > 1) {
            if ((rc = slab_after(s->comm, s->comp, s->ev_f[g]))) return rc;
            if (g == 0 && s->comm != s->comp) HIPCHK(hipStreamWaitEvent(s->comm, s->ev_rows_done, 0));   // the previous stage's row pass is done reading w4_recv
            if ((rc = slab_xchg(s, B.w4_send, B.w4_recv, 4 * fld, f0 * fld, (f1 - f0) * fld))) return rc;
        }
    }
    m->primed = 2;
    if (c->world > 1 && (rc = slab_after(s->comp, s->comm, s->ev_w4))) return rc;
    // row pass (main.cpp:154-237, y part) in row chunks; each chunk's tendency rows leave while the next chunk is computed
    const int rows = c->XL / s->nch;
    for (int h = 0; h < s->nch; ++h) {
        if ((rc = launch_row<ROW_FUSED>(c, fused_row_args(m, h * rows, rows)))) return rc;
        if (c->world > 1) {
            if ((rc = slab_after(s->comm, s->comp, s->ev_r[h]))) return rc;
            if (h == 0 && s->comm != s->comp) HIPCHK(hipStreamWaitEvent(s->comm, s->ev_fwd_done, 0));    // the previous forward pass is done with t_recv
            if ((rc = slab_xchg(s, B.t_send, B.t_recv, fld, (size_t)h * rows * c->grp[0].ncols, (size_t)rows * c->grp[0].ncols))) return rc;
        }
    }
    if (s->comm != s->comp) HIPCHK(hipEventRecord(s->ev_rows_done, s->comp));
    if (c->world > 1 && (rc = slab_after(s->comp, s->comm, s->ev_t))) return rc;
    // forward x pass, viscosity, mask, RK stage update, derivatives of the new stage state (main.cpp:148,179-212,237-251,296-312)
    if ((rc = model_col_fwd(m, stage))) return rc;
    if (s->comm != s->comp) HIPCHK(hipEventRecord(s->ev_fwd_done, s->comp));
    return FB_OK;
}

extern "C" int fb_slab_step(fb_slab *s, int nsteps)
{
    SLAB_READY(s);
    if (nsteps < 0) return fail(FB_EINVAL, "fb_slab_step: nsteps < 0");
    if (nsteps == 0) return FB_OK;
    int rc;
    if (!s->m->primed) {
        if ((rc = slab_prime(s))) return rc;
        HIPCHK(hipEventRecord(s->ev_rows_done, s->comp));
        HIPCHK(hipEventRecord(s->ev_fwd_done, s->comp));
    }
    if (s->ncg > 1 && s->m->primed == 1 && (rc = slab_groups_prologue(s))) return rc;
    for (int n = 0; n < nsteps; ++n)
        for (int k = 0; k < 4; ++k)                         // main.cpp:288-317
            if ((rc = slab_stage(s, k))) return rc;
    return FB_OK;
}

extern "C" int fb_slab_synchronize(fb_slab *s)
{
    if (!s) return fail(FB_EINVAL, "slab NULL");
    HIPCHK(hipStreamSynchronize(s->comp));
    HIPCHK(hipStreamSynchronize(s->comm));
    return FB_OK;
}

// hand-overs between the rank's compute stream and a host-side copy stream (record path and source staging of the driver)
extern "C" int fb_slab_record_event(fb_slab *s, void *event)
{
    if (!s || !event) return fail(FB_EINVAL, "fb_slab_record_event: NULL");
    HIPCHK(hipEventRecord((hipEvent_t)event, s->comp));
    return FB_OK;
}
extern "C" int fb_slab_wait_event(fb_slab *s, void *event)
{
    if (!s || !event) return fail(FB_EINVAL, "fb_slab_wait_event: NULL");
    HIPCHK(hipStreamWaitEvent(s->comp, (hipEvent_t)event, 0));
    return FB_OK;
}

// wall time of `nsteps` steps on this rank's compute stream (HIP events); the caller takes the maximum over ranks
extern "C" int fb_slab_time_steps(fb_slab *s, int nsteps, float *total_ms)
{
    SLAB_READY(s);
    if (!total_ms) return fail(FB_EINVAL, "fb_slab_time_steps: NULL");
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipEventRecord(e0, s->comp));
    int rc = fb_slab_step(s, nsteps);
    HIPCHK(hipEventRecord(e1, s->comp));
    HIPCHK(hipEventSynchronize(e1));
    HIPCHK(hipEventElapsedTime(total_ms, e0, e1));
    hipEventDestroy(e0); hipEventDestroy(e1);
    return rc;
}
