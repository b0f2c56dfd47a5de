// invert_pres.cpp -- nonlinear-balance pressure from psi on the MI355X engine.
//
// Mirror of the reference's invert_pres.cpp:65-192 (second consumer of the operator API,
// SURVEY.md section 8(f) rank 1): stdin lines "from=>to"; for each, read psi, r2c, the three second
// derivatives by chained gradx/grady (:139-145), in-place dealiase (:148-150), three c2r, Gaussian
// curvature product (:159), r2c, laplacian, rho*(f*lap(psi) + 2*curv) (:164-169), invertLaplacian,
// c2r, subtract the reference point (:182-185), write.  Options -x -y as the reference (:71-79),
// plus --npts --lx --ly for the compile-time constants of configuration.hpp.
// Everything pointwise runs through the C ABI's bit-exact float32 sweeps:
//   a*b - c*c                       == fb_jacobian(u=-a, v=c, dzdx=b, dzdy=c, src=NULL)
//   f*t + 2*l                       == fb_spec_evolve(l, l, 1) then fb_spec_axpy(., t, f)
#include <getopt.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "fftwfop_hip.hpp"

static void trim(char *str) { size_t n = strlen(str); while (n && (str[n - 1] == '\n' || str[n - 1] == '\r')) str[--n] = '\0'; }   // invert_pres.cpp:45-53

int main(int argc, char *args[])
{
    const float rho = 1.0f, f = 1e-5;                                  // configuration.hpp:10-11
    size_t ref_x = 0, ref_y = 0;
    int npts = 768; float LX = 600000.0f, LY = 600000.0f;
    static struct option lopts[] = {{"npts", 1, 0, 1}, {"lx", 1, 0, 2}, {"ly", 1, 0, 3}, {0, 0, 0, 0}};
    int opt;
    while ((opt = getopt_long(argc, args, "x:y:", lopts, NULL)) != EOF) {
        switch (opt) {
        case 'x': ref_x = atoi(optarg); break;
        case 'y': ref_y = atoi(optarg); break;
        case 1: npts = atoi(optarg); break;
        case 2: LX = (float)atof(optarg); break;
        case 3: LY = (float)atof(optarg); break;
        }
    }
    const int XPTS = npts, YPTS = npts;
    const size_t GRIDS = (size_t)XPTS * YPTS, HALF_GRIDS = (size_t)XPTS * (YPTS / 2 + 1);
    fb_ctx *fop = nullptr;
    fb_must(fb_create(&fop, XPTS, YPTS, LX, LY), "fb_create");

    auto ralloc = [&]() { return (float *)fbw_malloc(sizeof(float) * GRIDS); };
    auto calloc_ = [&]() { return (float *)fbw_malloc(sizeof(fftwf_complex) * HALF_GRIDS); };
    float *pres = ralloc(), *psi = ralloc(), *dpsidx2 = ralloc(), *dpsidy2 = ralloc(), *dpsidxdy = ralloc(), *gaus_curv = ralloc();   // :84-89
    float *tmp_c = calloc_(), *psi_c = calloc_(), *dpsidx2_c = calloc_(), *dpsidy2_c = calloc_(), *dpsidxdy_c = calloc_(), *lap_pres_c = calloc_();   // :92-97
    std::vector<float> host(GRIDS);

    char filename[1024], from_file[1024], to_file[1024];
    const char sep[] = "=>";
    while (fgets(filename, 1024, stdin) != NULL) {                       // :114
        trim(filename);
        char *sep_beg = strstr(filename, sep);
        if (sep_beg == NULL) { printf("Error reading input: %s. Continue next line...\n", filename); continue; }
        size_t l = sep_beg - filename;
        memcpy(from_file, filename, l); from_file[l] = '\0';
        strcpy(to_file, sep_beg + strlen(sep));

        fb_must(fb_read_field(from_file, host.data(), GRIDS), "readField");                  // :132
        fb_must(fb_memcpy_h2d(fop, psi, host.data(), sizeof(float) * GRIDS), "h2d");
        fb_must(fb_r2c(fop, psi, psi_c), "r2c");                                              // :135
        fb_must(fb_gradx(fop, psi_c, tmp_c), "gradx"); fb_must(fb_gradx(fop, tmp_c, dpsidx2_c), "gradx");   // :139-140
        fb_must(fb_grady(fop, psi_c, tmp_c), "grady"); fb_must(fb_grady(fop, tmp_c, dpsidy2_c), "grady");   // :142-143
        fb_must(fb_gradx(fop, tmp_c, dpsidxdy_c), "gradx");                                                // :145
        fb_must(fb_dealiase(fop, dpsidx2_c, dpsidx2_c), "dealiase");                          // :148-150 (in place)
        fb_must(fb_dealiase(fop, dpsidy2_c, dpsidy2_c), "dealiase");
        fb_must(fb_dealiase(fop, dpsidxdy_c, dpsidxdy_c), "dealiase");
        fb_must(fb_c2r(fop, dpsidx2_c, dpsidx2, 1), "c2r");                                   // :153-155
        fb_must(fb_c2r(fop, dpsidy2_c, dpsidy2, 1), "c2r");
        fb_must(fb_c2r(fop, dpsidxdy_c, dpsidxdy, 1), "c2r");
        fb_must(fb_negate(fop, dpsidx2), "negate");                                           // gaus_curv = dx2*dy2 - dxdy^2   :159
        fb_must(fb_jacobian(fop, dpsidx2, dpsidxdy, dpsidy2, dpsidxdy, nullptr, gaus_curv), "curvature");
        fb_must(fb_r2c(fop, gaus_curv, lap_pres_c), "r2c");                                   // :161
        fb_must(fb_laplacian(fop, psi_c, tmp_c), "laplacian");                                // :164
        fb_must(fb_spec_evolve(fop, lap_pres_c, lap_pres_c, 1.0f, lap_pres_c), "2*curv");     // :166-169 with rho = 1
        fb_must(fb_spec_axpy(fop, lap_pres_c, tmp_c, f), "f*lap(psi)");
        (void)rho;
        fb_must(fb_invert_laplacian(fop, lap_pres_c, tmp_c), "invertLaplacian");              // :171
        fb_must(fb_c2r(fop, tmp_c, pres, 1), "c2r");                                          // :172
        fb_must(fb_memcpy_d2h(fop, host.data(), pres, sizeof(float) * GRIDS), "d2h");
        const float ref_val = host[ref_x + (size_t)XPTS * ref_y];                             // :182-185
        for (size_t i = 0; i < GRIDS; ++i) host[i] -= ref_val;
        fb_must(fb_write_field(to_file, host.data(), GRIDS), "writeField");                   // :187
    }
    printf("Program ends. Congrats!\n");
    return 0;
}
