// shim_check.cpp -- reference-shaped host code against fftwfop_hip.hpp: one tendency evaluation
// written exactly like getDvortdt of main.cpp:146-244 (same call order, same operator names),
// on device buffers.  Prints max |dvortdt| and a checksum; tests/test_host_cpp.py compares with the oracle.
#include <cmath>
#include <cstdio>
#include <vector>

#include "fftwfop_hip.hpp"

const int XPTS = 256, YPTS = 256, GRIDS = XPTS * YPTS, HALF_GRIDS = XPTS * (YPTS / 2 + 1);
const float LX = 600000.0f, LY = 600000.0f, NU = 6.5f;
fftwf_operation<XPTS, YPTS> fop(LX, LY);                                    // main.cpp:33

int main(int argc, char **argv)
{
    const char *in = argc > 1 ? argv[1] : "input/initial_vorticity.bin";
    const char *outf = argc > 2 ? argv[2] : "dvortdt_c.bin";
    std::vector<float> h(GRIDS), hc(2 * HALF_GRIDS);
    fb_must(fb_read_field(in, h.data(), GRIDS), "readField");
    float *vort = (float *)fbw_malloc(sizeof(float) * GRIDS), *u = (float *)fbw_malloc(sizeof(float) * GRIDS),
          *v = (float *)fbw_malloc(sizeof(float) * GRIDS), *dvortdx = (float *)fbw_malloc(sizeof(float) * GRIDS),
          *dvortdy = (float *)fbw_malloc(sizeof(float) * GRIDS), *dvortdt = (float *)fbw_malloc(sizeof(float) * GRIDS);
    fftwf_complex *vort_c = (fftwf_complex *)fbw_malloc(sizeof(fftwf_complex) * HALF_GRIDS),
                  *lvort_c = (fftwf_complex *)fbw_malloc(sizeof(fftwf_complex) * HALF_GRIDS),
                  *dvortdt_c = (fftwf_complex *)fbw_malloc(sizeof(fftwf_complex) * HALF_GRIDS),
                  *tmp_c = (fftwf_complex *)fbw_malloc(sizeof(fftwf_complex) * HALF_GRIDS),
                  *psi_c = (fftwf_complex *)fbw_malloc(sizeof(fftwf_complex) * HALF_GRIDS);
    fbw_plan p_fwd_vort = fbw_plan_dft_r2c_2d(fop, vort, vort_c), p_fwd_dvortdt = fbw_plan_dft_r2c_2d(fop, dvortdt, dvortdt_c),
             p_bwd_dvortdx = fbw_plan_dft_c2r_2d(fop, tmp_c, dvortdx), p_bwd_dvortdy = fbw_plan_dft_c2r_2d(fop, tmp_c, dvortdy),
             p_bwd_u = fbw_plan_dft_c2r_2d(fop, tmp_c, u), p_bwd_v = fbw_plan_dft_c2r_2d(fop, tmp_c, v);
    fb_ctx *ctx = fop.handle();
    fb_must(fb_memcpy_h2d(ctx, vort, h.data(), sizeof(float) * GRIDS), "h2d");
    fbw_execute(p_fwd_vort);                                                                 // main.cpp:256
    fop.laplacian(vort_c, lvort_c);                                                          // :148
    fop.gradx(vort_c, tmp_c); fbw_execute(p_bwd_dvortdx); fb_must(fb_backward_normalize(ctx, dvortdx), "norm");   // :151-154
    fop.grady(vort_c, tmp_c); fbw_execute(p_bwd_dvortdy); fb_must(fb_backward_normalize(ctx, dvortdy), "norm");   // :165-168
    fop.invertLaplacian(vort_c, psi_c);                                                      // :179
    fop.grady(psi_c, tmp_c); fbw_execute(p_bwd_u); fb_must(fb_backward_normalize(ctx, u), "norm"); fb_must(fb_negate(ctx, u), "neg");   // :198-201
    fop.gradx(psi_c, tmp_c); fbw_execute(p_bwd_v); fb_must(fb_backward_normalize(ctx, v), "norm");                 // :212-214
    fb_must(fb_jacobian(ctx, u, v, dvortdx, dvortdy, nullptr, dvortdt), "jacobian");         // :225-227
    fbw_execute(p_fwd_dvortdt);                                                              // :237
    fb_must(fb_spec_axpy(ctx, (float *)dvortdt_c, (const float *)lvort_c, NU), "axpy");      // :240-243
    fop.dealiase(dvortdt_c, dvortdt_c);                                                      // :296 (in place)
    fb_must(fb_memcpy_d2h(ctx, hc.data(), dvortdt_c, sizeof(float) * 2 * HALF_GRIDS), "d2h");
    fb_must(fb_write_field(outf, hc.data(), 2 * HALF_GRIDS), "writeField");
    double s = 0; float mx = 0;
    for (float x : hc) { s += x; mx = std::fmax(mx, std::fabs(x)); }
    printf("rk1_c: max |.| = %.6e  sum = %.9e\n", mx, s);
    return 0;
}
