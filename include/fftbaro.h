/*
 * fftbaro.h -- C ABI of the MI355X-native pseudospectral barotropic-vorticity engine.
 *
 * Drop-in boundary for the hot path of meteorologytoday/XLab-FFTBarotropic: every entry
 * point below names the reference interface it replaces (paths relative to the reference
 * root).  Plain pointers and sizes only; no C++/torch types.  All `float *` arguments named
 * d_* are DEVICE pointers (HIP); spectra are interleaved (re,im) float32 in the reference's
 * half-spectrum layout HIDX(i,j) = (ny/2+1)*i + j (configuration.hpp:32), real fields are
 * IDX(i,j) = ny*i + j (configuration.hpp:31).
 *
 * Every function returns an int status (FB_OK == 0); the reference's methods are `void` with
 * no error path (fftwfop.hpp:20-24) -- see INTEGRATION.md for the binding a maintainer adds.
 * Work is enqueued on the context's HIP stream (default: the null stream); functions do not
 * synchronise unless stated.
 *
 * There is NO CPU fallback: when no HIP device is usable fb_create fails with FB_EHIP.
 */
#ifndef FFTBARO_H
#define FFTBARO_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FB_OK            0
#define FB_EINVAL        1   /* bad argument (null pointer, unsupported size, ...) */
#define FB_ENOMEM        2   /* host or device allocation failed                  */
#define FB_EHIP          3   /* HIP runtime error; see fb_last_error()            */
#define FB_EIO           4   /* file open / short read / short write              */
#define FB_EUNSUPPORTED  5   /* grid size not supported by the kernels            */

const char *fb_strerror(int status);
const char *fb_last_error(void);          /* thread-local detail of the last failure */
int fb_version(void);                      /* 100*major + minor                       */
/* one process per GPU: pick this process's (this thread's) device before creating a context; hipGetDeviceCount / hipSetDevice */
int fb_device_count(int *count);
int fb_set_device(int ordinal);
/* 1 if (nx,ny) is supported: each a power of two in [64, 16384] or 3*2^k in [192, 3072]
 * (the reference's default NPTS = 768, configuration.hpp:18) */
int fb_size_supported(int nx, int ny);

/* ---------------------------------------------------------------------------------------
 * Context = operator tables + FFT plans.
 * Replaces: fftwf_operation<XPTS,YPTS>::fftwf_operation(Lx,Ly) / ~fftwf_operation()
 *           (fftwfop.hpp:18-19, fftwfop.cpp:5-85) and the eight fftwf_plan_dft_{r2c,c2r}_2d
 *           calls of main.cpp:126-135.  Grid size is a run-time argument here
 *           (configuration.hpp:18-21 fixes it at compile time).
 * ------------------------------------------------------------------------------------- */
typedef struct fb_ctx fb_ctx;
int fb_create(fb_ctx **out, int nx, int ny, float lx, float ly);
int fb_destroy(fb_ctx *ctx);
int fb_set_stream(fb_ctx *ctx, void *hip_stream);     /* hipStream_t; NULL = null stream */
int fb_synchronize(fb_ctx *ctx);                       /* hipStreamSynchronize            */
/* copies the five coefficient tables to HOST buffers (any may be NULL): gradx_coe[nx],
 * grady_coe[ny/2+1], laplacian_coe / laplacian_coe_inverse / dealiasing_mask [nx*(ny/2+1)]
 * (fftwfop.cpp:15-68; the debug print of :26-36,70-77) */
int fb_get_tables(fb_ctx *ctx, float *gradx_coe, float *grady_coe, float *laplacian_coe,
                  float *laplacian_coe_inverse, float *dealiasing_mask);

/* ---------------------------------------------------------------------------------------
 * Buffers.  Replaces fftwf_malloc / fftwf_free (main.cpp:103-123) with device memory.
 * ------------------------------------------------------------------------------------- */
int fb_malloc(void **d_ptr, size_t bytes);
int fb_free(void *d_ptr);
int fb_memcpy_h2d(fb_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);  /* synchronous */
int fb_memcpy_d2h(fb_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);  /* synchronous */
int fb_memset0(fb_ctx *ctx, void *d_dst, size_t bytes);

/* Asynchronous record path (no reference counterpart: main.cpp:266-282 writes synchronously from its only
 * thread).  Pinned host buffers, non-blocking streams and events as opaque handles, so that a C/C++ host can
 * overlap the D2H copies and the file writes of a record step with the following RK4 steps:
 *   compute stream: fb_model_get_vort/diag -> fb_event_record(e1, compute)
 *   copy stream   : fb_stream_wait_event(copy, e1) -> fb_memcpy_d2h_async(copy, ...) -> fb_event_record(e2, copy)
 *   writer thread : fb_event_synchronize(e2) -> fb_write_field(...)                (host/barotropic_main.cpp) */
int fb_malloc_host(void **h_ptr, size_t bytes);
int fb_free_host(void *h_ptr);
int fb_stream_create(void **stream);                 /* hipStreamNonBlocking; pass to fb_set_stream */
int fb_stream_destroy(void *stream);
int fb_stream_synchronize(void *stream);
int fb_event_create(void **event);
int fb_event_destroy(void *event);
int fb_event_record(void *event, void *stream);
int fb_stream_wait_event(void *stream, void *event);
int fb_event_synchronize(void *event);
/* events that carry a time stamp, and the GPU time between two of them once both have fired: how the drop-in driver prices its
 * step loop (steps/s without the record steps' stalls) without synchronising it */
int fb_event_create_timing(void **event);
int fb_event_elapsed_ms(void *start, void *stop, float *ms);
int fb_memcpy_d2h_async(void *stream, void *h_dst, const void *d_src, size_t bytes);
/* the other direction, for the FIFO source (main-shallow-water.cpp:304: a reader thread fills a pinned buffer, the copy
 * stream carries it to the device, the compute stream waits for the copy's event before fb_model_set_source) */
int fb_memcpy_h2d_async(void *stream, void *d_dst, const void *h_src, size_t bytes);

/* ---------------------------------------------------------------------------------------
 * Spectral operators on device half spectra (nx*(ny/2+1) complex).  in == out is allowed
 * (main-shallow-water.cpp:327, invert_pres.cpp:148-150).
 * Replaces fftwf_operation::gradx/grady/laplacian/invertLaplacian/dealiase
 * (fftwfop.hpp:20-24, fftwfop.cpp:87-124).  Bit-exact with the reference's float32 forms.
 * ------------------------------------------------------------------------------------- */
int fb_gradx(fb_ctx *ctx, const float *d_in, float *d_out);
int fb_grady(fb_ctx *ctx, const float *d_in, float *d_out);
int fb_laplacian(fb_ctx *ctx, const float *d_in, float *d_out);
int fb_invert_laplacian(fb_ctx *ctx, const float *d_in, float *d_out);
int fb_dealiase(fb_ctx *ctx, const float *d_in, float *d_out);

/* ---------------------------------------------------------------------------------------
 * 2-D FFTs.  Replaces fftwf_execute() on the r2c / c2r plans (main.cpp:154,168,186,200,214,
 * 237,256,275).  Unnormalised, r2c sign -, c2r sign +; c2r accepts non-Hermitian input with
 * FFTW's semantics (SURVEY.md note N2) and -- unlike FFTW -- preserves its input, so the
 * copy_for_c2r dance of main.cpp:273,281 is unnecessary (harmless if kept).
 * fb_c2r with normalize != 0 also applies fftwf_backward_normalize (main.cpp:37-41).
 * ------------------------------------------------------------------------------------- */
int fb_r2c(fb_ctx *ctx, const float *d_in_real, float *d_out_spec);
int fb_c2r(fb_ctx *ctx, const float *d_in_spec, float *d_out_real, int normalize);

/* ---------------------------------------------------------------------------------------
 * Pointwise sweeps the reference driver does in its lambdas.
 * ------------------------------------------------------------------------------------- */
/* data[i] /= GRIDS                                   fftwf_backward_normalize, main.cpp:37-41 */
int fb_backward_normalize(fb_ctx *ctx, float *d_real);
/* data[i] = -data[i]                                 main.cpp:201                            */
int fb_negate(fb_ctx *ctx, float *d_real);
/* out = -u*dzdx - v*dzdy + src (src NULL = 0)        main.cpp:225-227                        */
int fb_jacobian(fb_ctx *ctx, const float *d_u, const float *d_v, const float *d_dzdx,
                const float *d_dzdy, const float *d_src, float *d_out);
/* acc += x * a   on spectra                          main.cpp:240-243 (viscous add)          */
int fb_spec_axpy(fb_ctx *ctx, float *d_acc, const float *d_x, float a);
/* out = base + rk * a  on spectra                    evolve(), main.cpp:246-251              */
int fb_spec_evolve(fb_ctx *ctx, const float *d_base, const float *d_rk, float a, float *d_out);
/* out = base + (k1 + 2 k2 + 2 k3 + k4) * dt / 6      main.cpp:309-312                        */
int fb_spec_rk4_combine(fb_ctx *ctx, const float *d_base, const float *d_k1, const float *d_k2,
                        const float *d_k3, const float *d_k4, float dt, float *d_out);

/* ---------------------------------------------------------------------------------------
 * Fused RK4 model: the hot loop of main.cpp:259-323 / main-shallow-water.cpp:277-338 with
 * the state resident in HBM.  One fb_model_step() == one iteration of the reference's step
 * loop body without the record path (a13-a15 of SURVEY.md section 8 fused into the FFT passes).
 * ------------------------------------------------------------------------------------- */
typedef struct fb_model fb_model;
int fb_model_create(fb_model **out, fb_ctx *ctx, float nu, float dt);      /* NU, dt: configuration.hpp:17,34 */
int fb_model_destroy(fb_model *m);
/* readField + fftwf_execute(p_fwd_vort)              main.cpp:143-144,256 */
int fb_model_set_vort(fb_model *m, const float *d_vort_real);
/* vort_src (device real field, copied; NULL = zeros) main.cpp:110,226; refreshed per step by
 * VortSrcRecipeReader::read in main-shallow-water.cpp:304 */
int fb_model_set_source(fb_model *m, const float *d_src_real);
int fb_model_step(fb_model *m, int nsteps);
/* enable != 0: fb_model_step replays one captured RK4 step as a hipGraph (16 kernel launches per step are
 * launch-bound on small grids).  Needs a non-null context stream (fb_set_stream); otherwise steps run eagerly. */
int fb_model_use_graph(fb_model *m, int enable);
/* record path: c2r of a copy of vort_c + normalise   main.cpp:273-281 */
int fb_model_get_vort(fb_model *m, float *d_vort_real);
/* stage-0 record dumps psi, u, v (any may be NULL)   main.cpp:181-222 */
int fb_model_get_diag(fb_model *m, float *d_psi, float *d_u, float *d_v);
/* vort_c in the reference layout */
int fb_model_get_spectrum(fb_model *m, float *d_spec);
int fb_model_set_spectrum(fb_model *m, const float *d_spec);
/* bytes of HBM the model holds, and the algorithmic bytes of one step (320*nx*ny) */
int fb_model_info(fb_model *m, size_t *hbm_bytes, size_t *alg_bytes_per_step);
/* times `nsteps` steps with HIP events on the model's stream;
 * total_ms = wall time of the whole batch of steps on the device */
int fb_model_time_steps(fb_model *m, int nsteps, float *total_ms);
/* the same steps with a HIP-event pair around every kernel launch (on the model's stream).
 * Kernel classes: 0 = k_col_strided<+1> (4 fields, backward x sub-pass), 1 = the fused row pass,
 * 2 = k_col_strided<-1> (tendency, forward x sub-pass), 3 = k_col_mid -- or, where the single-pass x
 * transform is in use (nx = 4096 on one GPU), k_col_full, and classes 0 and 2 have no launches.
 * ms_sum[4] receives the summed durations, launches[4] the launch counts. */
int fb_model_profile_steps(fb_model *m, int nsteps, float *ms_sum, int *launches);

/* ---------------------------------------------------------------------------------------
 * Multi-GPU: slab decomposition over the GPUs of one node, one process per GPU, RCCL all-to-all transposes over
 * xGMI between the row and the column passes.  No reference counterpart -- the reference is
 * single-process; one fb_slab_step() computes what one iteration of main.cpp:259-323 computes, on this rank's share.
 *
 * Rank r of `world` (a power of two) owns the x rows [r*XL, (r+1)*XL), XL = nx/world, of every physical field, and of
 * every spectral field KA "active" ky columns [r*KA, (r+1)*KA) -- the columns below world*KA hold every mode inside
 * the dealiasing circle and are exchanged twice per RK stage -- plus KF "frozen" columns [world*KA + r*KF, ...): their
 * modes are masked (fftwfop.cpp:57-61), never change (SURVEY.md note N1), and cross the links once per fb_slab_set_vort_local.
 * The engine owns the exchange buffers, two HIP streams (compute / communication) and the schedule: per stage the four
 * derivative fields leave field group by field group behind their backward x sub-pass, the tendency leaves row chunk
 * by row chunk behind the row pass (fb_slab_plan reports the granularity, which follows the message size).
 *
 * Transports: fb_slab_connect_rccl (the product: grouped ncclSend/ncclRecv; rank 0 obtains the id from
 * fb_slab_unique_id and hands it to the other ranks -- file, environment, torch.distributed, MPI ...),
 * fb_slab_connect_local (all ranks as threads of ONE process on ONE GPU: rehearsal of the schedule),
 * fb_slab_connect_callback (the caller moves the bytes).  world == 1 needs no transport.
 * ------------------------------------------------------------------------------------- */
typedef struct fb_slab fb_slab;
#define FB_UNIQUE_ID_BYTES 128
int fb_slab_unique_id(char *id128);                       /* ncclGetUniqueId */
int fb_slab_create(fb_slab **out, int nx, int ny, float lx, float ly, float nu, float dt, int rank, int world);
int fb_slab_destroy(fb_slab *s);
int fb_slab_connect_rccl(fb_slab *s, const char *id128);  /* ncclCommInitRank: collective over all ranks */
int fb_local_hub_create(void **hub, int world);
int fb_local_hub_destroy(void *hub);
int fb_slab_connect_local(fb_slab *s, void *hub);
/* For every peer p: `count` floats at send + p*stride + offset go to recv + rank*stride + offset of peer p (own block
 * included); must be complete, or ordered on hip_stream, when the callback returns.  Return 0 on success. */
typedef int (*fb_alltoall_fn)(void *user, const float *d_send, float *d_recv, size_t stride, size_t offset, size_t count, void *hip_stream);
int fb_slab_connect_callback(fb_slab *s, fb_alltoall_fn fn, void *user);
/* this rank's rows [XL][ny] (device pointers): readField + r2c (main.cpp:143-144,256), vort_src (main-shallow-water.cpp:304;
 * NULL = zeros), the record path (main.cpp:273-281) */
int fb_slab_set_vort_local(fb_slab *s, const float *d_rows);
int fb_slab_set_source_local(fb_slab *s, const float *d_rows);
int fb_slab_get_vort_local(fb_slab *s, float *d_rows);
/* the stage-0 record dumps psi, u, v (main.cpp:181-222) on this rank's rows; any may be NULL */
int fb_slab_get_diag_local(fb_slab *s, float *d_psi, float *d_u, float *d_v);
int fb_slab_step(fb_slab *s, int nsteps);
int fb_slab_synchronize(fb_slab *s);
/* the rank's compute stream is the engine's own: record an event behind what has been queued on it (fb_slab_get_*_local ->
 * copy stream) or make it wait for one (H2D of a new source -> fb_slab_set_source_local); events from fb_event_create */
int fb_slab_record_event(fb_slab *s, void *event);
int fb_slab_wait_event(fb_slab *s, void *event);
int fb_slab_time_steps(fb_slab *s, int nsteps, float *total_ms);
/* exchanges a known pattern of world*count floats through the connected transport; *wrong_words = 0 when every word arrived */
int fb_slab_transport_selftest(fb_slab *s, size_t count, size_t *wrong_words);
/* what is connected: transport name ("rccl", "local", "callback", "none" for world == 1), the size / rank / device of the transport's
 * own communicator (RCCL: ncclCommCount, ncclCommUserRank, ncclCommCuDevice; -1 where the transport has none) and this rank's HIP
 * device ordinal.  Any pointer may be NULL. */
int fb_slab_transport_info(fb_slab *s, char *name, size_t cap, int *comm_ranks, int *comm_rank, int *comm_device, int *hip_device);
int fb_slab_info(fb_slab *s, int *rows_local, int *cols_active, int *cols_frozen, int *ky0_active, int *ky0_frozen,
                 int *field_groups, int *row_chunks);
/* host logic, no GPU needed: XL, KA, KF of a decomposition */
int fb_slab_geometry(int nx, int ny, int world, int *rows_local, int *cols_active, int *cols_frozen);
/* host logic, no GPU needed: the column groups a rank's active columns are cut into (1, or 2 where a stage is pipelined by
 * column groups: BASELINE configs 4 and 5); cols2[g] = columns per rank of group g (0 for an absent group): a rank's active
 * slab [rank*KA, (rank+1)*KA) is cut locally, its first cols2[0] columns are group 0 */
int fb_slab_col_groups(int nx, int ny, int world, int *ngroups, int *cols2);
/* host logic, no GPU needed: the operations of ONE RK stage in issue order, ops[i] = 16*kind + argument, kind =
 * 1 backward x sub-pass of field group g, 2 all-to-all of the derivative fields (field group g; column group g when
 * pipelined by column groups), 3 row pass of row chunk h, 4 all-to-all of tendency chunk h (every column group), 5 forward x
 * pass + RK update (of column group g), 6 backward x sub-pass of all four fields of column group g.
 * Returns the number of operations (negative never; 0 on error). */
int fb_slab_plan(int nx, int ny, int world, int *field_groups, int *row_chunks, int *ops, int cap);
/* lower level: a context bound to one rank's slabs (the operator / FFT entry points above need world == 1) */
int fb_create_slab(fb_ctx **out, int nx, int ny, float lx, float ly, int rank, int world);

/* ---------------------------------------------------------------------------------------
 * Field I/O on HOST buffers.  Replaces writeField / readField (fieldio.hpp:5-6,
 * fieldio.cpp:7-33): identical bytes on disk and identical stderr lines, plus a status.
 * ------------------------------------------------------------------------------------- */
int fb_write_field(const char *filename, const float *h_data, size_t len);
int fb_read_field(const char *filename, float *h_data, size_t len);

/* ---------------------------------------------------------------------------------------
 * Initial-field synthesis on HOST buffers (nx*ny float32), the inputs of the benchmark configs.
 * kind = "elliptic"  makefield-elliptic-vortex.cpp:14-52
 *        "kuo2004"   makefield-Kuo2004.cpp:30-41 + field_generator.cpp:10-28 (buffer zeroed first)
 *        "gaussian"  makefield-gaussian.cpp:14-31
 *        "const"     makefield-const-vortex.cpp:14-38
 * fb_make_source_kuo2004: the FIFO producer's source cake, vort_src_input.cpp:35-46.
 * ------------------------------------------------------------------------------------- */
int fb_make_field(const char *kind, int nx, int ny, float lx, float ly, float *h_vort);
int fb_make_source_kuo2004(int nx, int ny, float lx, float ly, float duration, float *h_src);

#ifdef __cplusplus
}
#endif
#endif /* FFTBARO_H */
