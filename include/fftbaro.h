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
int fb_memcpy_d2h_async(void *stream, void *h_dst, const void *d_src, size_t bytes);

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
 * Slab decomposition over the GPUs of one node (one process per GPU).
 * No reference counterpart -- the reference is single-process.  Process `rank` of `world`
 * (a power of two) owns the x rows [rank*XL, (rank+1)*XL), XL = nx/world, of every physical
 * field and the ky columns [ky0, ky0+KS) of every spectral field (KS = 16*ceil((ny/2+1)/(16*world))).
 * The engine does the local passes; the CALLER does the two all-to-all transposes per RK stage
 * (RCCL via torch.distributed, or ncclSend/ncclRecv) on four buffers it owns, E = nx*KS complex:
 *   w4_send [world][4][XL][KS]   destination-blocked: block d = rows d*XL..(d+1)*XL of the four fields,
 *                                 written by FB_PH_PRIME / FB_PH_COL_FWD, finished by FB_PH_COL_BWD
 *   w4_recv [world][4][XL][KS]   block s comes from rank s (its ky slab of this rank's rows)
 *   t_send  [world][XL][KS]      written by FB_PH_ROW / FB_PH_R2C_ROWS; block d goes to rank d
 *   t_recv  [nx][KS]             block s (rows s*XL..) comes from rank s
 * so each transpose is ONE equal-split all-to-all of a contiguous buffer (4E and E complex).
 * One RK4 step = for stage in 0..3 { COL_BWD; all-to-all(w4); ROW; all-to-all(t); COL_FWD(stage) },
 * preceded once by PRIME after the state was set.  The record path (C2R_*) runs the transposes in
 * the opposite roles: C2R_COLS leaves [dst][XL][KS] in t_recv, all-to-all(t_recv -> t_send), C2R_ROWS.
 * ------------------------------------------------------------------------------------- */
int fb_create_slab(fb_ctx **out, int nx, int ny, float lx, float ly, int rank, int world);
int fb_slab_geometry(fb_ctx *ctx, int *rows_local, int *cols_per_slab, int *ky0, size_t *elems_per_field);
int fb_model_create_slab(fb_model **out, fb_ctx *ctx, float nu, float dt, float *d_w4_send, float *d_w4_recv,
                         float *d_t_send, float *d_t_recv);
#define FB_PH_PRIME     0   /* derivatives of the current vort_c -> w4_send                               */
#define FB_PH_COL_BWD   1   /* backward x sub-pass on w4_send (then transpose w4_send -> w4_recv)          */
#define FB_PH_ROW       2   /* w4_recv -> c2r rows, Jacobian (+ local rows of vort_src), r2c rows -> t_send */
#define FB_PH_COL_FWD   3   /* t_recv -> forward x pass, viscosity, mask, RK stage `stage`, derivatives    */
#define FB_PH_R2C_ROWS  4   /* d_real_in (local rows [XL][ny]) -> t_send           (set_vort, first half)  */
#define FB_PH_R2C_COLS  5   /* t_recv -> vort_c                                    (set_vort, second half) */
#define FB_PH_C2R_COLS  6   /* copy of vort_c -> x-backward-transformed columns in t_recv (get_vort, 1st half) */
#define FB_PH_C2R_ROWS  7   /* t_send ([src][XL][KS]) -> d_real_out (local rows, normalised)                  */
int fb_model_phase(fb_model *m, int phase, int stage, const float *d_real_in, float *d_real_out);

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
