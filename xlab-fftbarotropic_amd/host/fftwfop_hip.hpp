// fftwfop_hip.hpp -- header-only C++ mirror of the reference's operator class on top of the C ABI.
//
// `fftwf_operation<XPTS,YPTS>` keeps the reference's template signature and method names
// (fftwfop.hpp:9-29) so that reference-shaped host code compiles unchanged; every method
// forwards to include/fftbaro.h.  Buffers are device memory (fb_malloc / fbw_malloc, asynchronous on the
// context's stream) or the pinned host memory fftwf_malloc of include/fftw3_fb.h returns; with
// -DFFTWFOP_HIP_SYNCHRONOUS every method returns only when its result is readable by the host, which is what
// reference-shaped code that loops over the arrays between operator calls needs (fftw_shape_check.cpp).
// The FFTW entry points the drivers use (main.cpp:103-135,154) exist twice: with FFTW's exact names and
// signatures in include/fftw3_fb.h (lib/libfftw3f_fb.so), and as the asynchronous device-memory `fbw_*` forms below.
#ifndef FFTWFOP_HIP_HPP
#define FFTWFOP_HIP_HPP
#include <cassert>
#include <cstdio>
#include <cstdlib>

#include "../../include/fftbaro.h"

#ifndef FFTW3_FB_H
typedef float fftwf_complex[2];                       // same layout as FFTW's (fftw3.h)
#endif

inline void fb_must(int status, const char *what)
{
    if (status != FB_OK) {
        std::fprintf(stderr, "%s: %s (%s)\n", what, fb_strerror(status), fb_last_error());
        std::exit(1);                                  // the reference has no error path (fftwfop.hpp:20-24)
    }
}

template <int XPTS, int YPTS> class fftwf_operation {
private:
    fb_ctx *ctx;
#ifdef FFTWFOP_HIP_SYNCHRONOUS
    void done() { fb_must(fb_synchronize(ctx), "fftwf_operation"); }
#else
    void done() {}
#endif
    const int HALF_XPTS = (int)(XPTS / 2) + 1, HALF_YPTS = (int)(YPTS / 2) + 1, HALF_GRIDS = XPTS * HALF_YPTS;
public:
    fftwf_operation(float Lx, float Ly) : ctx(nullptr) { fb_must(fb_create(&ctx, XPTS, YPTS, Lx, Ly), "fftwf_operation"); }   // fftwfop.cpp:5-79
    ~fftwf_operation() { fb_destroy(ctx); }                                                                                   // fftwfop.cpp:81-85
    fftwf_operation(const fftwf_operation &) = delete;
    fftwf_operation &operator=(const fftwf_operation &) = delete;

    void gradx(fftwf_complex *in, fftwf_complex *out) { fb_must(fb_gradx(ctx, (const float *)in, (float *)out), "gradx"); done(); }                           // :87-94
    void grady(fftwf_complex *in, fftwf_complex *out) { fb_must(fb_grady(ctx, (const float *)in, (float *)out), "grady"); done(); }                           // :96-103
    void laplacian(fftwf_complex *in, fftwf_complex *out) { fb_must(fb_laplacian(ctx, (const float *)in, (float *)out), "laplacian"); done(); }               // :105-110
    void invertLaplacian(fftwf_complex *in, fftwf_complex *out) { fb_must(fb_invert_laplacian(ctx, (const float *)in, (float *)out), "invertLaplacian"); done(); }   // :112-117
    void dealiase(fftwf_complex *in, fftwf_complex *out) { fb_must(fb_dealiase(ctx, (const float *)in, (float *)out), "dealiase"); done(); }                  // :119-124

    inline int reflectedXWavenumberIndex(int i) { assert(i >= 1 && "Input of ReflectedXWavenumberIndex must >= 1"); return XPTS - i; }
    inline int HIDX(int i, int j) { return HALF_YPTS * i + j; }
    inline int R_HIDX(int i, int j) { return HIDX(this->reflectedXWavenumberIndex(i), j); }

    fb_ctx *handle() { return ctx; }
};

// ---- what the drivers take from FFTW, on device memory -----------------------------------------
inline void *fbw_malloc(size_t bytes) { void *p = nullptr; fb_must(fb_malloc(&p, bytes), "fbw_malloc"); return p; }   // fftwf_malloc, main.cpp:103-123
inline void fbw_free(void *p) { fb_free(p); }                                                                          // fftwf_free

// fftwf_plan_dft_r2c_2d / fftwf_plan_dft_c2r_2d (main.cpp:126-135): a plan binds two buffers
struct fbw_plan_s { fb_ctx *ctx; int c2r; float *real; fftwf_complex *spec; };
typedef fbw_plan_s *fbw_plan;
template <int XPTS, int YPTS>
inline fbw_plan fbw_plan_dft_r2c_2d(fftwf_operation<XPTS, YPTS> &fop, float *in, fftwf_complex *out) { return new fbw_plan_s{fop.handle(), 0, in, out}; }
template <int XPTS, int YPTS>
inline fbw_plan fbw_plan_dft_c2r_2d(fftwf_operation<XPTS, YPTS> &fop, fftwf_complex *in, float *out) { return new fbw_plan_s{fop.handle(), 1, out, in}; }
inline void fbw_execute(const fbw_plan p)                                                                              // fftwf_execute, main.cpp:154,...
{
    if (p->c2r) fb_must(fb_c2r(p->ctx, (const float *)p->spec, p->real, 0), "fbw_execute(c2r)");
    else fb_must(fb_r2c(p->ctx, p->real, (float *)p->spec), "fbw_execute(r2c)");
}
inline void fbw_destroy_plan(fbw_plan p) { delete p; }
#endif
