/*
 * fftw3_fb.h -- the subset of FFTW3's single-precision API that XLab-FFTBarotropic uses, with FFTW's exact names
 * and signatures, implemented on the MI355X engine (lib/libfftw3f_fb.so -> libfftbaro.so).
 *
 * Replaces, for the reference's drivers, <fftw3.h> + -lfftw3f:
 *   fftwf_malloc / fftwf_free                         main.cpp:103-123, invert_pres.cpp:88-98
 *   fftwf_plan_dft_r2c_2d(n0,n1,in,out,flags)         main.cpp:126-127, main-shallow-water.cpp:139-140
 *   fftwf_plan_dft_c2r_2d(n0,n1,in,out,flags)         main.cpp:129-135, main-shallow-water.cpp:142-148
 *   fftwf_execute(plan)                               main.cpp:154,168,186,200,214,237,256,275
 *   fftwf_destroy_plan / fftwf_cleanup                (never called by the reference; provided for completeness)
 * Semantics are FFTW's: n0 = XPTS is the slow dimension, n1 = YPTS the fast one; out of r2c / in of c2r are
 * n0*(n1/2+1) fftwf_complex; transforms are unnormalised (r2c sign -1, c2r sign +1); c2r accepts non-Hermitian
 * input the way FFTW does (complex inverse DFT along n0, then per row a 1-D c2r that ignores the imaginary parts
 * at j = 0 and j = n1/2 -- SURVEY.md note N2).  Differences: c2r PRESERVES its input (FFTW destroys it, hence
 * copy_for_c2r in main.cpp:273-281 -- harmless); `flags` is accepted and ignored (planning never touches the
 * arrays, as with FFTW_ESTIMATE); sizes must satisfy fb_size_supported() or the planner returns NULL as FFTW does
 * on failure.
 *
 * Memory model: fftwf_malloc returns PINNED HOST memory that the GPU addresses directly, so reference-shaped host
 * code (readField into the buffer, host loops over it, main.cpp:37-41,143,225-227) keeps working unchanged;
 * fftwf_execute is synchronous like FFTW's.  Buffers obtained from fb_malloc (device memory) are accepted too and
 * are the fast path; the fused RK4 model (fb_model_*) is the product's hot path, this shim is the drop-in surface.
 */
#ifndef FFTW3_FB_H
#define FFTW3_FB_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef float fftwf_complex[2];
typedef struct fftwf_plan_s *fftwf_plan;

#define FFTW_MEASURE (0U)
#define FFTW_DESTROY_INPUT (1U << 0)
#define FFTW_PRESERVE_INPUT (1U << 4)
#define FFTW_ESTIMATE (1U << 6)

void *fftwf_malloc(size_t n);
void fftwf_free(void *p);
float *fftwf_alloc_real(size_t n);
fftwf_complex *fftwf_alloc_complex(size_t n);
fftwf_plan fftwf_plan_dft_r2c_2d(int n0, int n1, float *in, fftwf_complex *out, unsigned flags);
fftwf_plan fftwf_plan_dft_c2r_2d(int n0, int n1, fftwf_complex *in, float *out, unsigned flags);
void fftwf_execute(const fftwf_plan p);
void fftwf_destroy_plan(fftwf_plan p);
void fftwf_cleanup(void);

#ifdef __cplusplus
}
#endif
#endif /* FFTW3_FB_H */
