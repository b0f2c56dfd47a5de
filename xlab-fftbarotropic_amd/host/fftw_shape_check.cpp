// fftw_shape_check.cpp -- a translation unit written against the reference's OWN interfaces, with the call lines
// of main.cpp (fftwf_malloc :103-123, fftwf_plan_dft_*_2d(XPTS, YPTS, in, out, FFTW_ESTIMATE) :126-135,
// fftwf_execute :154-256, readField/writeField :143-144,268, fop.gradx(...) :151, host loops :37-41,201,225-243):
// it includes <fftw3.h>, "fieldio.hpp" and the operator class, and links -lfftw3f_fb -lfieldio instead of
// -lfftw3f.  One tendency evaluation + dealiase (rk1_c) is written out; tests/test_host_cpp.py compares it with
// the oracle.  Everything here runs on fftwf_malloc'ed (pinned host) buffers the way the reference's code does.
#include <cmath>
#include <cstdio>
#include <fftw3.h>

#include "fieldio.hpp"
#include "fftwfop_hip.hpp"

const int XPTS = 256, YPTS = 256, GRIDS = XPTS * YPTS, HALF_YPTS = YPTS / 2 + 1, HALF_GRIDS = XPTS * HALF_YPTS;
const float LX = 600000.0f, LY = 600000.0f, NU = 6.5f;

float *vort, *u, *v, *dvortdx, *dvortdy, *dvortdt;
fftwf_plan p_fwd_vort, p_bwd_dvortdx, p_bwd_dvortdy, p_bwd_u, p_bwd_v, p_fwd_dvortdt;
fftwf_complex *vort_c, *lvort_c, *dvortdt_c, *tmp_c, *psi_c, *rk1_c;

fftwf_operation<XPTS, YPTS> fop(LX, LY);                                   // main.cpp:33

void fftwf_backward_normalize(float *data)                                 // main.cpp:37-41
{
    for (int i = 0; i < GRIDS; ++i) data[i] /= GRIDS;
}

int main(int argc, char *args[])
{
    const char *in = argc > 1 ? args[1] : "input/initial_vorticity.bin";
    const char *outf = argc > 2 ? args[2] : "rk1_c.bin";
    vort      = (float*) fftwf_malloc(sizeof(float) * GRIDS);
    u         = (float*) fftwf_malloc(sizeof(float) * GRIDS);
    v         = (float*) fftwf_malloc(sizeof(float) * GRIDS);
    dvortdx   = (float*) fftwf_malloc(sizeof(float) * GRIDS);
    dvortdy   = (float*) fftwf_malloc(sizeof(float) * GRIDS);
    dvortdt   = (float*) fftwf_malloc(sizeof(float) * GRIDS);
    vort_c    = (fftwf_complex*) fftwf_malloc(sizeof(fftwf_complex) * HALF_GRIDS);
    lvort_c   = (fftwf_complex*) fftwf_malloc(sizeof(fftwf_complex) * HALF_GRIDS);
    dvortdt_c = (fftwf_complex*) fftwf_malloc(sizeof(fftwf_complex) * HALF_GRIDS);
    tmp_c     = (fftwf_complex*) fftwf_malloc(sizeof(fftwf_complex) * HALF_GRIDS);
    psi_c     = (fftwf_complex*) fftwf_malloc(sizeof(fftwf_complex) * HALF_GRIDS);
    rk1_c     = (fftwf_complex*) fftwf_malloc(sizeof(fftwf_complex) * HALF_GRIDS);

    p_fwd_vort       = fftwf_plan_dft_r2c_2d(XPTS, YPTS, vort, vort_c, FFTW_ESTIMATE);
    p_fwd_dvortdt    = fftwf_plan_dft_r2c_2d(XPTS, YPTS, dvortdt, dvortdt_c, FFTW_ESTIMATE);
    p_bwd_dvortdx    = fftwf_plan_dft_c2r_2d(XPTS, YPTS, tmp_c, dvortdx, FFTW_ESTIMATE);
    p_bwd_dvortdy    = fftwf_plan_dft_c2r_2d(XPTS, YPTS, tmp_c, dvortdy, FFTW_ESTIMATE);
    p_bwd_u          = fftwf_plan_dft_c2r_2d(XPTS, YPTS, tmp_c, u, FFTW_ESTIMATE);
    p_bwd_v          = fftwf_plan_dft_c2r_2d(XPTS, YPTS, tmp_c, v, FFTW_ESTIMATE);
    if (!p_fwd_vort || !p_fwd_dvortdt || !p_bwd_dvortdx || !p_bwd_dvortdy || !p_bwd_u || !p_bwd_v) return 1;

    readField(in, vort, GRIDS);                                            // main.cpp:143-144
    fftwf_execute(p_fwd_vort);                                             // :256

    fop.laplacian(vort_c, lvort_c);                                        // :148
    fop.gradx(vort_c, tmp_c);                                              // :151
    fftwf_execute(p_bwd_dvortdx); fftwf_backward_normalize(dvortdx);       // :154
    fop.grady(vort_c, tmp_c);                                              // :165
    fftwf_execute(p_bwd_dvortdy); fftwf_backward_normalize(dvortdy);       // :168
    fop.invertLaplacian(vort_c, psi_c);                                    // :179
    fop.grady(psi_c, tmp_c);                                               // :198
    fftwf_execute(p_bwd_u); fftwf_backward_normalize(u);                   // :200
    for (int i = 0; i < GRIDS; ++i) u[i] = -u[i];                           // :201
    fop.gradx(psi_c, tmp_c);                                               // :212
    fftwf_execute(p_bwd_v); fftwf_backward_normalize(v);                   // :214
    for (int i = 0; i < GRIDS; ++i) dvortdt[i] = - u[i] * dvortdx[i] - v[i] * dvortdy[i];   // :225-227 (vort_src = 0)
    fftwf_execute(p_fwd_dvortdt);                                          // :237
    for (int i = 0; i < HALF_GRIDS; ++i) {                                  // :240-243
        dvortdt_c[i][0] += lvort_c[i][0] * NU;
        dvortdt_c[i][1] += lvort_c[i][1] * NU;
    }
    fop.dealiase(dvortdt_c, rk1_c);                                        // :296

    writeField(outf, (float *)rk1_c, 2 * (size_t)HALF_GRIDS);
    double s = 0; float mx = 0;
    for (int i = 0; i < HALF_GRIDS; ++i) for (int c = 0; c < 2; ++c) { s += rk1_c[i][c]; mx = std::fmax(mx, std::fabs(rk1_c[i][c])); }
    printf("rk1_c: max |.| = %.6e  sum = %.9e\n", mx, s);
    fftwf_destroy_plan(p_fwd_vort); fftwf_free(vort);
    return 0;
}
