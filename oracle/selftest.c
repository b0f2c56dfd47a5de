/* selftest.c -- exercises every entry point of the CPU oracle (fb_oracle.c) on small grids, for the
 * sanitizer build:  make -C oracle asan  ->  oracle/_asan/selftest  (gcc -fsanitize=address,undefined).
 * TEST INFRASTRUCTURE ONLY (like the oracle itself).  Prints a few checksums; exit status 0 = every
 * internal consistency check passed.  Run by tests/test_oracle.py::test_oracle_under_asan_ubsan. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fb_oracle.h"

static double sum(const float *a, size_t n) { double s = 0; for (size_t i = 0; i < n; ++i) s += a[i]; return s; }

static int one_grid(int nx, int ny, int steps)
{
    const size_t grids = (size_t)nx * ny, half = (size_t)nx * (ny / 2 + 1);
    float *v = malloc(sizeof(float) * grids), *w = malloc(sizeof(float) * grids), *psi = malloc(sizeof(float) * grids);
    float *u = malloc(sizeof(float) * grids), *src = malloc(sizeof(float) * grids);
    float *c = malloc(sizeof(float) * 2 * half), *d = malloc(sizeof(float) * 2 * half);
    if (!v || !w || !psi || !u || !src || !c || !d) return 1;
    int bad = 0;
    fbo_make_elliptic(nx, ny, 6e5f, 6e5f, v);
    fbo_make_gaussian(nx, ny, 6e5f, 6e5f, w);
    fbo_make_const_vortex(nx, ny, 6e5f, 6e5f, w);
    fbo_make_kuo2004(nx, ny, 6e5f, 6e5f, w);
    memset(src, 0, sizeof(float) * grids);
    fbo_add_cake_kuo2004(nx, ny, 6e5f, 6e5f, src, 3.5e5f, 3e5f, 3e-3f / 10800.0f, 3e4f);
    /* FFT round trip: c2r(r2c(x)) / GRIDS == x */
    fbo_r2c_2d(nx, ny, v, c);
    fbo_c2r_2d(nx, ny, c, w);
    double e = 0, n2 = 0;
    for (size_t i = 0; i < grids; ++i) { double x = w[i] / (double)grids - v[i]; e += x * x; n2 += (double)v[i] * v[i]; }
    if (!(sqrt(e / n2) < 1e-6)) { fprintf(stderr, "round trip %dx%d: %g\n", nx, ny, sqrt(e / n2)); bad = 1; }
    /* operators, in place and out of place */
    fbo_op *op = fbo_op_create(nx, ny, 6e5f, 6e5f);
    fbo_gradx(op, c, d); fbo_grady(op, d, d); fbo_laplacian(op, c, d); fbo_invert_laplacian(op, d, d); fbo_dealiase(op, d, d);
    fbo_op_destroy(op);
    float *line = malloc(sizeof(float) * 2 * nx);
    for (int i = 0; i < 2 * nx; ++i) line[i] = (float)(i % 7) - 3.0f;
    fbo_fft1d(nx, -1, line); fbo_fft1d(nx, +1, line);
    free(line);
    /* model: a few forced steps, record path, spectrum round trip */
    fbo_model *m = fbo_model_create(nx, ny, 6e5f, 6e5f, 6.5f, 3.0f);
    fbo_model_set_vort(m, v);
    fbo_model_set_source(m, src);
    const double mean0 = sum(v, grids);
    for (int s = 0; s < steps; ++s) fbo_model_step(m);
    fbo_model_get_diag(m, psi, u, w);
    fbo_model_get_spectrum(m, c);
    fbo_model_set_spectrum(m, c);
    fbo_model_set_source(m, NULL);
    fbo_model_step(m);
    fbo_model_get_vort(m, w);
    for (size_t i = 0; i < grids; ++i) if (!isfinite(w[i])) { bad = 1; break; }
    printf("%dx%d: sum(vort0) = %.9e  sum(vort after %d steps) = %.9e  max|u| = %.4f\n", nx, ny, mean0, steps + 1, sum(w, grids),
           (double)fabsf(u[grids / 2 + ny / 3]));
    fbo_model_destroy(m);
    /* field I/O round trip */
    char path[] = "/tmp/fbo_selftest_XXXXXX";
    int fd = mkstemp(path);
    if (fd >= 0) {
        if (fbo_write_field(path, v, grids) != 0 || fbo_read_field(path, w, grids) != 0 || memcmp(v, w, sizeof(float) * grids)) bad = 1;
        FILE *f = fopen(path, "wb");                       /* FIFO protocol: flag 0, flag 1 + field, EOF */
        if (f) {
            fputc(0, f); fputc(1, f); fwrite(v, sizeof(float), grids, f); fclose(f);
            f = fopen(path, "rb");
            (void)fbo_fifo_read(f, w, grids); (void)fbo_fifo_read(f, w, grids); (void)fbo_fifo_read(f, w, grids);
            fclose(f);
            if (memcmp(v, w, sizeof(float) * grids)) bad = 1;
        }
        remove(path);
    }
    free(v); free(w); free(psi); free(u); free(src); free(c); free(d);
    return bad;
}

int main(void)
{
    int bad = 0;
    bad |= one_grid(64, 64, 3);
    bad |= one_grid(96, 96, 2);          /* 3 * 2^k */
    bad |= one_grid(128, 64, 2);         /* non-square */
    printf(bad ? "selftest FAILED\n" : "selftest ok\n");
    return bad;
}
