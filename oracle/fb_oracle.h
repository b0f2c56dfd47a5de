/*
 * fb_oracle.h -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * A plain-C restatement of the reference's RK4 hot path
 * (meteorologytoday/XLab-FFTBarotropic), kept in the reference's own unfused loop
 * structure so that it doubles as the timed CPU baseline ("port").
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library -- and there only as the checker.  The product path (HIP, behind
 * include/fftbaro.h) never calls into it.
 *
 * PARITY PIN STATUS
 *   - Operator tables / pointwise operators / driver loop: restated line by line from the
 *     reference sources cited on each function.
 *   - The 2-D FFTs live in FFTW3 single precision (">= 3.3.4", README.md:16; un-vendored,
 *     unpinned), which is absent from this image, so the reference's FFT-dependent programs
 *     are UNBUILDABLE here and the reference ships no fixtures: for r2c/c2r and everything
 *     downstream the oracle is "parity unpinned" by the reference.  It is pinned instead to
 *     the mathematical DFT definition (fp64 numpy rfft2/irfft2 with FFTW's documented
 *     conventions) and to analytic known answers; see tests/test_oracle.py.
 *   - fieldio + initial-field generators: pinned bit-for-bit against oracle/_ref (the
 *     reference's own fieldio.cpp / makefield-*.cpp compiled unmodified; they need no FFTW).
 */
#ifndef FB_ORACLE_H
#define FB_ORACLE_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- operator tables + pointwise spectral operators: fftwfop.cpp:5-124 ---- */
typedef struct fbo_op {
    int nx, ny, hy;          /* XPTS, YPTS, HALF_YPTS = ny/2+1                */
    float lx, ly;
    float *gradx_coe;        /* [nx]      fftwfop.cpp:15-20                    */
    float *grady_coe;        /* [hy]      fftwfop.cpp:22-24                    */
    float *lap;              /* [nx*hy]   fftwfop.cpp:40-54                    */
    float *lap_inv;          /* [nx*hy]   same, (0,0) := 1.0f (:42-43)          */
    float *mask;             /* [nx*hy]   fftwfop.cpp:57-68                    */
} fbo_op;

fbo_op *fbo_op_create(int nx, int ny, float lx, float ly);
void    fbo_op_destroy(fbo_op *op);
/* in/out: interleaved (re,im) float32, nx*hy complex, HIDX(i,j)=hy*i+j; in==out allowed */
void fbo_gradx(const fbo_op *op, const float *in, float *out);           /* :87-94   */
void fbo_grady(const fbo_op *op, const float *in, float *out);           /* :96-103  */
void fbo_laplacian(const fbo_op *op, const float *in, float *out);       /* :105-110 */
void fbo_invert_laplacian(const fbo_op *op, const float *in, float *out);/* :112-117 */
void fbo_dealiase(const fbo_op *op, const float *in, float *out);        /* :119-124 */

/* ---- 2-D FFTs with FFTW's conventions (main.cpp:126-135): unnormalised, r2c sign -,
 *      c2r sign +, half spectrum [nx][ny/2+1]; c2r = complex inverse DFT along x for every
 *      ky column, then 1-D c2r along y that ignores Im at j=0 and j=ny/2.
 *      fbo_c2r_2d does not modify its input (FFTW's c2r may; callers must not rely on it). */
void fbo_r2c_2d(int nx, int ny, const float *in, float *out_c);
void fbo_c2r_2d(int nx, int ny, const float *in_c, float *out);
/* 1-D complex FFT, n with prime factors {2,3,5}; sign = -1 forward / +1 inverse; in place */
void fbo_fft1d(int n, int sign, float *data /* n complex */);

/* ---- RK4 model: main.cpp:103-123 (buffers), :146-251 (tendency/evolve), :259-317 (loop) */
typedef struct fbo_model fbo_model;
fbo_model *fbo_model_create(int nx, int ny, float lx, float ly, float nu, float dt);
void  fbo_model_destroy(fbo_model *m);
void  fbo_model_set_vort(fbo_model *m, const float *vort);      /* readField + r2c, :143-144,256 */
void  fbo_model_set_source(fbo_model *m, const float *src);     /* vort_src (NULL = zeros)      */
void  fbo_model_step(fbo_model *m);                              /* one RK4 step, :286-317        */
void  fbo_model_get_vort(fbo_model *m, float *vort);            /* record path, :273-281         */
/* stage-0 diagnostics as the record path dumps them (main.cpp:181-222): psi, u, v of the
 * current state; any pointer may be NULL */
void  fbo_model_get_diag(fbo_model *m, float *psi, float *u, float *v);
void  fbo_model_get_spectrum(fbo_model *m, float *vort_c);      /* copy of vort_c                */
void  fbo_model_set_spectrum(fbo_model *m, const float *vort_c);
/* main-shallow-water.cpp variant is arithmetically identical; the only
 * difference on the path is vort_src being refreshed per step by the caller. */

/* ---- initial-field generators (inputs of BASELINE.json configs) ---- */
void fbo_make_elliptic(int nx, int ny, float lx, float ly, float *vort);   /* makefield-elliptic-vortex.cpp:14-52 */
void fbo_make_kuo2004(int nx, int ny, float lx, float ly, float *vort);    /* makefield-Kuo2004.cpp:30-41 (zeroed) */
void fbo_make_gaussian(int nx, int ny, float lx, float ly, float *vort);   /* makefield-gaussian.cpp:14-31 */
void fbo_make_const_vortex(int nx, int ny, float lx, float ly, float *vort); /* makefield-const-vortex.cpp:14-38 */
void fbo_add_cake_kuo2004(int nx, int ny, float lx, float ly, float *data,
                          float cx, float cy, float zeta0, float scale_r); /* field_generator.cpp:10-28 */

/* ---- field I/O: fieldio.cpp:7-33 (same bytes on disk; adds a status) ---- */
int fbo_write_field(const char *filename, const float *data, size_t len);
int fbo_read_field(const char *filename, float *data, size_t len);

/* ---- FIFO vorticity-source protocol: vorticity_source.cpp:112-133 ----
 * returns 0 = ok (flag 0 or new field read), 1 = EOF on flag, 2 = short field read */
int fbo_fifo_read(void *FILE_ptr, float *vort_src, size_t grids);

#ifdef __cplusplus
}
#endif
#endif
