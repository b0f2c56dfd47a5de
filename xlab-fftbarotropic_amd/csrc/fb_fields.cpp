// fb_fields.cpp -- host-side initial-field synthesis (C ABI), the inputs of BASELINE.json's
// configs.  Restates the reference generators with run-time grid size; the float/double
// promotions of every sub-expression are the reference's (C++ usual arithmetic conversions:
// float variables, double literals, pow(float,int) -> double).
#include <cmath>
#include <cstring>
#include <string>

#include "../../include/fftbaro.h"

namespace {

// radius lambda of makefield-elliptic-vortex.cpp:21, makefield-gaussian.cpp:21
inline float radius(float x, float y, float cx, float cy)
{
    return sqrtf(std::pow(x - cx, 2) + std::pow(y - cy, 2));
}

// field_generator.cpp:10-28 ("cake" of Kuo 2004); note the reference walks j<XPTS for y and
// i<YPTS for x, which only matters for non-square grids
void add_cake(int nx, int ny, float lx, float ly, float *data, float cx, float cy, float zeta_0, float scale_r)
{
    const float DX = lx / nx, DY = ly / ny;                         // configuration.hpp:23-24
    for (size_t j = 0; j < (size_t)nx; ++j) {
        const float y = j * DY;
        for (size_t i = 0; i < (size_t)ny; ++i) {
            const float x = i * DX;
            const float r = sqrtf(std::pow(x - cx, 2.0) + std::pow(y - cy, 2.0)) / scale_r;   // field_generator.cpp:6-8,21
            if (r < 1) data[(size_t)ny * i + j] += zeta_0 * (1 - exp(-30.0 / r * exp(1.0 / (r - 1.0))));   // :24
        }
    }
}

}  // namespace

extern "C" int fb_make_field(const char *kind, int nx, int ny, float lx, float ly, float *vort)
{
    if (!kind || !vort || nx <= 0 || ny <= 0) return FB_EINVAL;
    const std::string k(kind);
    const float dx = lx / nx, dy = ly / ny;
    if (k == "elliptic") {                                            // makefield-elliptic-vortex.cpp:14-52
        const float centerx = lx / 2.0, centery = ly / 2.0, epsilon = 0.7, lambda = 2.0, zeta0 = .005f,
                    r_i = 30000.0, r_o = 60000.0;
        for (int i = 0; i < nx; ++i) {
            const float x = i * dx;
            for (int j = 0; j < ny; ++j) {
                const float y = j * dy;
                const float r = radius(x, y, centerx, centery);
                float c;
                if (r == 0.0f) c = 0; else c = (y - centery) / r;     // :24-29
                const float alpha = sqrtf((1.0 - std::pow(epsilon, 2)) / (1.0 - std::pow(epsilon * c, 2)));   // :30
                const float r_i_alpha = r_i * alpha, r_o_alpha = r_o * alpha;
                float &out = vort[(size_t)ny * i + j];
                if (r <= r_i_alpha) out = zeta0;
                else if (r <= r_o_alpha) {
                    const float r_prime = (r - r_i_alpha) / (r_o_alpha - r_i_alpha);
                    out = zeta0 * (1.0 - exp(-lambda / r_prime * exp(1.0 / (r_prime - 1))));   // :46
                } else out = 0;
            }
        }
    } else if (k == "kuo2004") {                                      // makefield-Kuo2004.cpp:30-41, buffer zeroed first
        memset(vort, 0, sizeof(float) * (size_t)nx * ny);
        add_cake(nx, ny, lx, ly, vort, lx / 2.0, ly / 2.0, 1.5e-2, 10000.0);
        add_cake(nx, ny, lx, ly, vort, lx / 2.0 + 50000.0, ly / 2.0, 3e-3, 30000.0);
    } else if (k == "gaussian") {                                     // makefield-gaussian.cpp:14-31
        const float centerx = lx / 2.0, centery = ly / 2.0, zeta0 = 1e-3;
        for (int i = 0; i < nx; ++i) {
            const float x = i * dx;
            for (int j = 0; j < ny; ++j) {
                const float y = j * dy;
                const float r = radius(x, y, centerx, centery);
                vort[(size_t)ny * i + j] = zeta0 * exp(-std::pow(r / 60000.0, 2.0));
            }
        }
    } else if (k == "const") {                                        // makefield-const-vortex.cpp:14-38
        const float centerx = lx / 2.0, centery = ly / 2.0, r_bound = 6000.0, zeta0 = 2e-5;
        for (int i = 0; i < nx; ++i) {
            const float x = i * dx;
            for (int j = 0; j < ny; ++j) {
                const float y = j * dy;
                vort[(size_t)ny * i + j] = radius(x, y, centerx, centery) <= r_bound ? zeta0 : 0;
            }
        }
    } else {
        return FB_EINVAL;
    }
    return FB_OK;
}

// the FIFO producer's source field, vort_src_input.cpp:35-46: one cake of 3e-3/duration
extern "C" int fb_make_source_kuo2004(int nx, int ny, float lx, float ly, float duration, float *src)
{
    if (!src || nx <= 0 || ny <= 0 || !(duration > 0.f)) return FB_EINVAL;
    memset(src, 0, sizeof(float) * (size_t)nx * ny);
    add_cake(nx, ny, lx, ly, src, lx / 2.0 + 50000.0, ly / 2.0, 3e-3 / duration, 30000.0);
    return FB_OK;
}
