/*
 * fb_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).  See fb_oracle.h.
 *
 * Every function cites the reference file:line it restates (paths relative to the
 * reference repository root).  Arithmetic is IEEE float32 except where the reference
 * itself promotes to double (pow(float,int) in the table constructor); compile with
 * -ffp-contract=off so that no FMA is formed the reference build (-O3, baseline x86-64,
 * Makefile:2) would not form either.
 */
#include "fb_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <errno.h>

#define HIDX(op, i, j) ((size_t)(op)->hy * (size_t)(i) + (size_t)(j))

/* fftwfop.hpp:7  const float TWOPI = (acos(-1.0f) * 2.0f);  -> 6.2831855f */
static float fbo_twopi(void) { return (float)(acosf(-1.0f) * 2.0f); }

/* ------------------------------------------------------------------------------------ */
/* operator tables: fftwfop.cpp:5-79                                                     */
/* ------------------------------------------------------------------------------------ */
fbo_op *fbo_op_create(int nx, int ny, float lx, float ly)
{
    fbo_op *op = (fbo_op *)calloc(1, sizeof(fbo_op));
    const float TWOPI = fbo_twopi();
    const int hx = nx / 2 + 1, hy = ny / 2 + 1;      /* fftwfop.hpp:13-15 */
    op->nx = nx; op->ny = ny; op->hy = hy; op->lx = lx; op->ly = ly;
    op->gradx_coe = (float *)malloc(sizeof(float) * (size_t)nx);
    op->grady_coe = (float *)malloc(sizeof(float) * (size_t)hy);
    op->lap       = (float *)malloc(sizeof(float) * (size_t)nx * hy);
    op->lap_inv   = (float *)malloc(sizeof(float) * (size_t)nx * hy);
    op->mask      = (float *)malloc(sizeof(float) * (size_t)nx * hy);

    /* fftwfop.cpp:11-12  (int) ceil(((float)XPTS)/3.0) : float promoted to double */
    const int dealiase_xw = (int)ceil(((double)(float)nx) / 3.0);
    const int dealiase_yw = (int)ceil(((double)(float)ny) / 3.0);

    /* fftwfop.cpp:15-20 */
    for (int i = 0; i < hx; ++i) op->gradx_coe[i] = TWOPI * ((float)i) / lx;
    for (int i = hx; i < nx; ++i) op->gradx_coe[i] = -op->gradx_coe[nx - i];
    /* fftwfop.cpp:22-24 */
    for (int j = 0; j < hy; ++j) op->grady_coe[j] = TWOPI * ((float)j) / ly;

    /* fftwfop.cpp:40-47: pow(float,int) promotes both to double; sum in double; the
     * negated double is rounded once when stored into the float table */
    for (int i = 0; i < hx; ++i) {
        for (int j = 0; j < hy; ++j) {
            double gx = (double)op->gradx_coe[i], gy = (double)op->grady_coe[j];
            double v = -(gx * gx + gy * gy);
            op->lap_inv[HIDX(op, i, j)] = (float)v;
            if (i == 0 && j == 0) op->lap_inv[HIDX(op, i, j)] = 1.0f;
            op->lap[HIDX(op, i, j)] = (float)v;
        }
    }
    /* fftwfop.cpp:49-54: upper half mirrored from N-i */
    for (int i = hx; i < nx; ++i) {
        for (int j = 0; j < hy; ++j) {
            op->lap_inv[HIDX(op, i, j)] = op->lap_inv[HIDX(op, nx - i, j)];
            op->lap[HIDX(op, i, j)]     = op->lap[HIDX(op, nx - i, j)];
        }
    }
    /* fftwfop.cpp:57-68: float generalized_wavenumber_square = pow(int,2)+pow(int,2) */
    const float gws = (float)((double)dealiase_xw * (double)dealiase_xw +
                              (double)dealiase_yw * (double)dealiase_yw);
    for (int i = 0; i < hx; ++i)
        for (int j = 0; j < hy; ++j)
            op->mask[HIDX(op, i, j)] =
                (((double)i * (double)i + (double)j * (double)j) >= (double)gws) ? 0.0f : 1.0f;
    for (int i = hx; i < nx; ++i)
        for (int j = 0; j < hy; ++j)
            op->mask[HIDX(op, i, j)] = op->mask[HIDX(op, nx - i, j)];
    return op;
}

void fbo_op_destroy(fbo_op *op)
{
    if (!op) return;
    free(op->gradx_coe); free(op->grady_coe); free(op->lap); free(op->lap_inv); free(op->mask);
    free(op);
}

/* fftwfop.cpp:87-94 */
void fbo_gradx(const fbo_op *op, const float *in, float *out)
{
    for (int i = 0; i < op->nx; ++i) {
        const float c = op->gradx_coe[i];
        for (int j = 0; j < op->hy; ++j) {
            size_t k = HIDX(op, i, j);
            float re = in[2 * k], im = in[2 * k + 1];
            out[2 * k]     = -im * c;
            out[2 * k + 1] = re * c;
        }
    }
}
/* fftwfop.cpp:96-103 */
void fbo_grady(const fbo_op *op, const float *in, float *out)
{
    for (int i = 0; i < op->nx; ++i) {
        for (int j = 0; j < op->hy; ++j) {
            const float c = op->grady_coe[j];
            size_t k = HIDX(op, i, j);
            float re = in[2 * k], im = in[2 * k + 1];
            out[2 * k]     = -im * c;
            out[2 * k + 1] = re * c;
        }
    }
}
/* fftwfop.cpp:105-110 */
void fbo_laplacian(const fbo_op *op, const float *in, float *out)
{
    size_t n = (size_t)op->nx * op->hy;
    for (size_t k = 0; k < n; ++k) {
        out[2 * k]     = in[2 * k] * op->lap[k];
        out[2 * k + 1] = in[2 * k + 1] * op->lap[k];
    }
}
/* fftwfop.cpp:112-117: a true division */
void fbo_invert_laplacian(const fbo_op *op, const float *in, float *out)
{
    size_t n = (size_t)op->nx * op->hy;
    for (size_t k = 0; k < n; ++k) {
        out[2 * k]     = in[2 * k] / op->lap_inv[k];
        out[2 * k + 1] = in[2 * k + 1] / op->lap_inv[k];
    }
}
/* fftwfop.cpp:119-124 */
void fbo_dealiase(const fbo_op *op, const float *in, float *out)
{
    size_t n = (size_t)op->nx * op->hy;
    for (size_t k = 0; k < n; ++k) {
        out[2 * k]     = in[2 * k] * op->mask[k];
        out[2 * k + 1] = in[2 * k + 1] * op->mask[k];
    }
}

/* ------------------------------------------------------------------------------------ */
/* FFT.  The reference takes these from FFTW3f (main.cpp:126-135,154,...).  FFTW is not    */
/* in the image; this is the oracle's own float32 mixed-radix Cooley-Tukey with twiddles  */
/* computed in double and rounded once (as FFTW's trig tables are).  Conventions are       */
/* FFTW's documented ones: forward = exp(-2 pi i jk/n), backward = exp(+...), unnormalised.*/
/* ------------------------------------------------------------------------------------ */
typedef struct { float re, im; } cpx;

typedef struct fbo_plan1d {
    int n;
    cpx *tw_fwd;   /* exp(-2 pi i k/n), k=0..n-1 */
    cpx *tw_bwd;   /* conj                         */
    struct fbo_plan1d *next;
} fbo_plan1d;

static fbo_plan1d *g_plans = NULL;

static fbo_plan1d *plan_get(int n)
{
    fbo_plan1d *p;
#pragma omp critical(fbo_plan)
    {
        for (p = g_plans; p; p = p->next) if (p->n == n) break;
        if (!p) {
            p = (fbo_plan1d *)malloc(sizeof(*p));
            p->n = n;
            p->tw_fwd = (cpx *)malloc(sizeof(cpx) * (size_t)n);
            p->tw_bwd = (cpx *)malloc(sizeof(cpx) * (size_t)n);
            for (int k = 0; k < n; ++k) {
                double a = -2.0 * M_PI * (double)k / (double)n;
                p->tw_fwd[k].re = (float)cos(a); p->tw_fwd[k].im = (float)sin(a);
                p->tw_bwd[k].re = p->tw_fwd[k].re; p->tw_bwd[k].im = -p->tw_fwd[k].im;
            }
            p->next = g_plans; g_plans = p;
        }
    }
    return p;
}

static inline cpx cmul(cpx a, cpx b)
{
    cpx r; r.re = a.re * b.re - a.im * b.im; r.im = a.re * b.im + a.im * b.re; return r;
}
static inline cpx cadd(cpx a, cpx b) { cpx r = { a.re + b.re, a.im + b.im }; return r; }
static inline cpx csub(cpx a, cpx b) { cpx r = { a.re - b.re, a.im - b.im }; return r; }

/* out[0..n) = DFT of in[0], in[stride], ...; tw = root table of size nroot, tstep = nroot/n */
static void fft_rec(int n, const cpx *in, int stride, cpx *out, const cpx *tw, int nroot, int sign)
{
    if (n == 1) { out[0] = in[0]; return; }
    int p;
    if (n % 4 == 0) p = 4; else if (n % 2 == 0) p = 2; else if (n % 3 == 0) p = 3;
    else if (n % 5 == 0) p = 5; else p = n;   /* prime: direct DFT */
    const int m = n / p;
    const int tstep = nroot / n;
    for (int q = 0; q < p; ++q)
        fft_rec(m, in + (size_t)q * stride, stride * p, out + (size_t)q * m, tw, nroot, sign);

    if (p == 2) {
        for (int k = 0; k < m; ++k) {
            cpx a = out[k], b = cmul(out[k + m], tw[(size_t)k * tstep]);
            out[k] = cadd(a, b); out[k + m] = csub(a, b);
        }
    } else if (p == 4) {
        for (int k = 0; k < m; ++k) {
            cpx a = out[k];
            cpx b = cmul(out[k + m],     tw[(size_t)k * tstep]);
            cpx c = cmul(out[k + 2 * m], tw[(size_t)2 * k * tstep]);
            cpx d = cmul(out[k + 3 * m], tw[(size_t)3 * k * tstep]);
            cpx s0 = cadd(a, c), s1 = csub(a, c), s2 = cadd(b, d), s3 = csub(b, d);
            /* multiply s3 by -i (forward) or +i (backward) */
            cpx t; if (sign < 0) { t.re = s3.im; t.im = -s3.re; } else { t.re = -s3.im; t.im = s3.re; }
            out[k]         = cadd(s0, s2);
            out[k + m]     = cadd(s1, t);
            out[k + 2 * m] = csub(s0, s2);
            out[k + 3 * m] = csub(s1, t);
        }
    } else {
        /* generic radix-p butterfly */
        cpx tmp[16]; cpx *t = tmp; cpx *heap = NULL;
        if (p > 16) { heap = (cpx *)malloc(sizeof(cpx) * (size_t)p); t = heap; }
        for (int k = 0; k < m; ++k) {
            for (int q = 0; q < p; ++q)
                t[q] = cmul(out[k + (size_t)q * m], tw[((size_t)q * k * tstep) % nroot]);
            for (int r = 0; r < p; ++r) {
                cpx acc = t[0];
                for (int q = 1; q < p; ++q)
                    acc = cadd(acc, cmul(t[q], tw[((size_t)q * r * m * tstep) % nroot]));
                out[k + (size_t)r * m] = acc;
            }
        }
        free(heap);
    }
}

static void fft1d_oop(const fbo_plan1d *pl, int sign, const cpx *in, cpx *out)
{
    fft_rec(pl->n, in, 1, out, sign < 0 ? pl->tw_fwd : pl->tw_bwd, pl->n, sign);
}

void fbo_fft1d(int n, int sign, float *data)
{
    fbo_plan1d *pl = plan_get(n);
    cpx *tmp = (cpx *)malloc(sizeof(cpx) * (size_t)n);
    fft1d_oop(pl, sign, (const cpx *)data, tmp);
    memcpy(data, tmp, sizeof(cpx) * (size_t)n);
    free(tmp);
}

#define COLBLK 8   /* columns gathered per block in the x pass */

/* x pass (complex DFT of length nx along the slow index) on a [nx][hy] half spectrum */
static void xpass(int nx, int hy, int sign, cpx *c)
{
    fbo_plan1d *pl = plan_get(nx);
#pragma omp parallel
    {
        cpx *bin  = (cpx *)malloc(sizeof(cpx) * (size_t)nx * COLBLK);
        cpx *bout = (cpx *)malloc(sizeof(cpx) * (size_t)nx);
#pragma omp for schedule(static)
        for (int j0 = 0; j0 < hy; j0 += COLBLK) {
            int nb = hy - j0 < COLBLK ? hy - j0 : COLBLK;
            for (int i = 0; i < nx; ++i)
                for (int b = 0; b < nb; ++b) bin[(size_t)b * nx + i] = c[(size_t)i * hy + j0 + b];
            for (int b = 0; b < nb; ++b) {
                fft1d_oop(pl, sign, bin + (size_t)b * nx, bout);
                for (int i = 0; i < nx; ++i) c[(size_t)i * hy + j0 + b] = bout[i];
            }
        }
        free(bin); free(bout);
    }
}

/* main.cpp:126-127,237,256  fftwf_plan_dft_r2c_2d + execute */
void fbo_r2c_2d(int nx, int ny, const float *in, float *out_c)
{
    const int hy = ny / 2 + 1;
    cpx *c = (cpx *)out_c;
    fbo_plan1d *pl = plan_get(ny);
    /* y pass: two real rows per complex FFT, z = row_a + i row_b */
#pragma omp parallel
    {
        cpx *z = (cpx *)malloc(sizeof(cpx) * (size_t)ny);
        cpx *Z = (cpx *)malloc(sizeof(cpx) * (size_t)ny);
#pragma omp for schedule(static)
        for (int i = 0; i < nx; i += 2) {
            const int two = (i + 1 < nx);
            const float *ra = in + (size_t)i * ny;
            const float *rb = two ? in + (size_t)(i + 1) * ny : NULL;
            for (int j = 0; j < ny; ++j) { z[j].re = ra[j]; z[j].im = two ? rb[j] : 0.0f; }
            fft1d_oop(pl, -1, z, Z);
            for (int j = 0; j < hy; ++j) {
                cpx p = Z[j], q = Z[(ny - j) % ny];
                /* A = (Z[k] + conj Z[n-k])/2 ; B = (Z[k] - conj Z[n-k])/(2i) */
                cpx A = { 0.5f * (p.re + q.re), 0.5f * (p.im - q.im) };
                cpx B = { 0.5f * (p.im + q.im), 0.5f * (q.re - p.re) };
                c[(size_t)i * hy + j] = A;
                if (two) c[(size_t)(i + 1) * hy + j] = B;
            }
        }
        free(z); free(Z);
    }
    xpass(nx, hy, -1, c);
}

/* main.cpp:129-135,154,168,200,214  fftwf_plan_dft_c2r_2d + execute (SURVEY N2 semantics) */
void fbo_c2r_2d(int nx, int ny, const float *in_c, float *out)
{
    const int hy = ny / 2 + 1;
    cpx *c = (cpx *)malloc(sizeof(cpx) * (size_t)nx * hy);
    memcpy(c, in_c, sizeof(cpx) * (size_t)nx * hy);
    xpass(nx, hy, +1, c);
    fbo_plan1d *pl = plan_get(ny);
#pragma omp parallel
    {
        cpx *Z = (cpx *)malloc(sizeof(cpx) * (size_t)ny);
        cpx *z = (cpx *)malloc(sizeof(cpx) * (size_t)ny);
#pragma omp for schedule(static)
        for (int i = 0; i < nx; i += 2) {
            const int two = (i + 1 < nx);
            const cpx *A = c + (size_t)i * hy;
            const cpx *B = two ? c + (size_t)(i + 1) * hy : NULL;
            /* Z = A_ext + i B_ext with Hermitian extension; Im at j=0 and j=ny/2 ignored */
            for (int j = 0; j < hy; ++j) {
                cpx a = A[j]; cpx b; if (two) b = B[j]; else { b.re = 0; b.im = 0; }
                if (j == 0 || 2 * j == ny) { a.im = 0.0f; b.im = 0.0f; }
                Z[j].re = a.re - b.im; Z[j].im = a.im + b.re;
                if (j != 0 && 2 * j != ny) { Z[ny - j].re = a.re + b.im; Z[ny - j].im = -a.im + b.re; }
            }
            fft1d_oop(pl, +1, Z, z);
            float *ra = out + (size_t)i * ny;
            float *rb = two ? out + (size_t)(i + 1) * ny : NULL;
            for (int j = 0; j < ny; ++j) { ra[j] = z[j].re; if (two) rb[j] = z[j].im; }
        }
        free(Z); free(z);
    }
    free(c);
}

/* ------------------------------------------------------------------------------------ */
/* RK4 model: main.cpp                                                                    */
/* ------------------------------------------------------------------------------------ */
struct fbo_model {
    int nx, ny, hy; size_t grids, half_grids;
    float nu, dt;
    fbo_op *fop;
    /* main.cpp:103-110 */
    float *vort, *u, *v, *dvortdx, *dvortdy, *dvortdt, *workspace, *vort_src;
    /* main.cpp:113-123 */
    float *vort_c0, *vort_c, *lvort_c, *dvortdt_c, *tmp_c, *psi_c, *rk1_c, *rk2_c, *rk3_c, *rk4_c;
};

static float *falloc(size_t n) { return (float *)calloc(n, sizeof(float)); }

fbo_model *fbo_model_create(int nx, int ny, float lx, float ly, float nu, float dt)
{
    fbo_model *m = (fbo_model *)calloc(1, sizeof(*m));
    m->nx = nx; m->ny = ny; m->hy = ny / 2 + 1;
    m->grids = (size_t)nx * ny; m->half_grids = (size_t)nx * m->hy;
    m->nu = nu; m->dt = dt;
    m->fop = fbo_op_create(nx, ny, lx, ly);
    m->vort = falloc(m->grids); m->u = falloc(m->grids); m->v = falloc(m->grids);
    m->dvortdx = falloc(m->grids); m->dvortdy = falloc(m->grids); m->dvortdt = falloc(m->grids);
    m->workspace = falloc(m->grids);
    m->vort_src = falloc(m->grids);      /* main.cpp:110 leaves it uninitialised; defined as zeros */
    size_t hc = 2 * m->half_grids;
    m->vort_c0 = falloc(hc); m->vort_c = falloc(hc); m->lvort_c = falloc(hc); m->dvortdt_c = falloc(hc);
    m->tmp_c = falloc(hc); m->psi_c = falloc(hc);
    m->rk1_c = falloc(hc); m->rk2_c = falloc(hc); m->rk3_c = falloc(hc); m->rk4_c = falloc(hc);
    return m;
}

void fbo_model_destroy(fbo_model *m)
{
    if (!m) return;
    fbo_op_destroy(m->fop);
    free(m->vort); free(m->u); free(m->v); free(m->dvortdx); free(m->dvortdy); free(m->dvortdt);
    free(m->workspace); free(m->vort_src);
    free(m->vort_c0); free(m->vort_c); free(m->lvort_c); free(m->dvortdt_c); free(m->tmp_c);
    free(m->psi_c); free(m->rk1_c); free(m->rk2_c); free(m->rk3_c); free(m->rk4_c);
    free(m);
}

/* main.cpp:37-41 */
static void backward_normalize(float *data, size_t grids)
{
    const float g = (float)(int)grids;           /* data[i] /= GRIDS  (int -> float) */
    for (size_t i = 0; i < grids; ++i) data[i] /= g;
}

void fbo_model_set_vort(fbo_model *m, const float *vort)
{
    memcpy(m->vort, vort, sizeof(float) * m->grids);          /* main.cpp:143-144 */
    fbo_r2c_2d(m->nx, m->ny, m->vort, m->vort_c);              /* main.cpp:256     */
}

void fbo_model_set_source(fbo_model *m, const float *src)
{
    if (src) memcpy(m->vort_src, src, sizeof(float) * m->grids);
    else memset(m->vort_src, 0, sizeof(float) * m->grids);
}

/* main.cpp:146-244 (debug dumps excluded) */
static void get_dvortdt(fbo_model *m)
{
    const size_t G = m->grids, H = m->half_grids;
    fbo_laplacian(m->fop, m->vort_c, m->lvort_c);                                  /* :148 */
    fbo_gradx(m->fop, m->vort_c, m->tmp_c);                                        /* :151 */
    fbo_c2r_2d(m->nx, m->ny, m->tmp_c, m->dvortdx); backward_normalize(m->dvortdx, G); /* :154 */
    fbo_grady(m->fop, m->vort_c, m->tmp_c);                                        /* :165 */
    fbo_c2r_2d(m->nx, m->ny, m->tmp_c, m->dvortdy); backward_normalize(m->dvortdy, G); /* :168 */
    fbo_invert_laplacian(m->fop, m->vort_c, m->psi_c);                             /* :179 */
    fbo_grady(m->fop, m->psi_c, m->tmp_c);                                         /* :198 */
    fbo_c2r_2d(m->nx, m->ny, m->tmp_c, m->u); backward_normalize(m->u, G);         /* :200 */
    for (size_t i = 0; i < G; ++i) m->u[i] = -m->u[i];                             /* :201 */
    fbo_gradx(m->fop, m->psi_c, m->tmp_c);                                         /* :212 */
    fbo_c2r_2d(m->nx, m->ny, m->tmp_c, m->v); backward_normalize(m->v, G);         /* :214 */
    for (size_t i = 0; i < G; ++i)                                                 /* :225-227 */
        m->dvortdt[i] = -m->u[i] * m->dvortdx[i] - m->v[i] * m->dvortdy[i] + m->vort_src[i];
    fbo_r2c_2d(m->nx, m->ny, m->dvortdt, m->dvortdt_c);                            /* :237 */
    for (size_t i = 0; i < H; ++i) {                                               /* :240-243 */
        m->dvortdt_c[2 * i]     += m->lvort_c[2 * i] * m->nu;
        m->dvortdt_c[2 * i + 1] += m->lvort_c[2 * i + 1] * m->nu;
    }
}

/* main.cpp:246-251 */
static void evolve(fbo_model *m, const float *rk, float dt)
{
    for (size_t i = 0; i < m->half_grids; ++i) {
        m->vort_c[2 * i]     = m->vort_c0[2 * i] + rk[2 * i] * dt;
        m->vort_c[2 * i + 1] = m->vort_c0[2 * i + 1] + rk[2 * i + 1] * dt;
    }
}

/* main.cpp:286-317 */
void fbo_model_step(fbo_model *m)
{
    const float dt = m->dt;
    memcpy(m->vort_c0, m->vort_c, sizeof(float) * 2 * m->half_grids);              /* :286 */
    for (int k = 0; k < 4; ++k) {
        get_dvortdt(m);                                                            /* :290 */
        switch (k) {
        case 0: fbo_dealiase(m->fop, m->dvortdt_c, m->rk1_c); evolve(m, m->rk1_c, dt / 2.0f); break; /* :296 */
        case 1: fbo_dealiase(m->fop, m->dvortdt_c, m->rk2_c); evolve(m, m->rk2_c, dt / 2.0f); break; /* :299 */
        case 2: fbo_dealiase(m->fop, m->dvortdt_c, m->rk3_c); evolve(m, m->rk3_c, dt); break;        /* :302 */
        case 3:
            fbo_dealiase(m->fop, m->dvortdt_c, m->rk4_c);                          /* :306 */
            for (size_t i = 0; i < 2 * m->half_grids; ++i)                         /* :309-312 */
                m->vort_c[i] = m->vort_c0[i] +
                    (m->rk1_c[i] + 2.0f * m->rk2_c[i] + 2.0f * m->rk3_c[i] + m->rk4_c[i]) * dt / 6.0f;
            break;
        }
    }
}

/* main.cpp:273-281 */
void fbo_model_get_vort(fbo_model *m, float *vort)
{
    fbo_c2r_2d(m->nx, m->ny, m->vort_c, m->vort); backward_normalize(m->vort, m->grids);
    memcpy(vort, m->vort, sizeof(float) * m->grids);
}

/* main.cpp:179-222 (the record-step dumps of stage 0) */
void fbo_model_get_diag(fbo_model *m, float *psi, float *u, float *v)
{
    const size_t G = m->grids;
    fbo_invert_laplacian(m->fop, m->vort_c, m->psi_c);
    if (psi) { fbo_c2r_2d(m->nx, m->ny, m->psi_c, m->workspace); backward_normalize(m->workspace, G);
               memcpy(psi, m->workspace, sizeof(float) * G); }
    if (u) { fbo_grady(m->fop, m->psi_c, m->tmp_c); fbo_c2r_2d(m->nx, m->ny, m->tmp_c, m->u);
             backward_normalize(m->u, G); for (size_t i = 0; i < G; ++i) m->u[i] = -m->u[i];
             memcpy(u, m->u, sizeof(float) * G); }
    if (v) { fbo_gradx(m->fop, m->psi_c, m->tmp_c); fbo_c2r_2d(m->nx, m->ny, m->tmp_c, m->v);
             backward_normalize(m->v, G); memcpy(v, m->v, sizeof(float) * G); }
}

void fbo_model_get_spectrum(fbo_model *m, float *vort_c) { memcpy(vort_c, m->vort_c, sizeof(float) * 2 * m->half_grids); }
void fbo_model_set_spectrum(fbo_model *m, const float *vort_c) { memcpy(m->vort_c, vort_c, sizeof(float) * 2 * m->half_grids); }

/* ------------------------------------------------------------------------------------ */
/* initial-field generators                                                               */
/* ------------------------------------------------------------------------------------ */
/* makefield-elliptic-vortex.cpp:14-52.  The reference mixes float variables with double
 * literals and pow(); each sub-expression below keeps the reference's promotion. */
void fbo_make_elliptic(int nx, int ny, float lx, float ly, float *vort)
{
    float centerx = lx / 2.0, centery = ly / 2.0, epsilon = 0.7, lambda = 2.0, zeta0 = .005f,
          r_i = 30000.0, r_o = 60000.0;
    float dx = lx / nx, dy = ly / ny;
    float r, r_i_alpha, r_o_alpha, r_prime;
    for (int i = 0; i < nx; ++i) {
        float x = i * dx;
        for (int j = 0; j < ny; ++j) {
            float y = j * dy;
            /* radius(): sqrtf(pow(x-centerx,2) + pow(y-centery,2)) -> double sum -> float arg */
            r = sqrtf((float)(pow((double)(x - centerx), 2) + pow((double)(y - centery), 2)));
            /* alpha() :22-31 */
            float c;
            if (r == 0.0f) c = 0; else c = (y - centery) / r;
            float alpha = sqrtf((float)((1.0 - pow((double)epsilon, 2)) /
                                        (1.0 - pow((double)(epsilon * c), 2))));
            r_i_alpha = r_i * alpha;
            r_o_alpha = r_o * alpha;
            size_t k = (size_t)ny * i + j;
            if (r <= r_i_alpha) {
                vort[k] = zeta0;
            } else if (r <= r_o_alpha) {
                r_prime = (r - r_i_alpha) / (r_o_alpha - r_i_alpha);
                /* zeta0 * (1.0 - exp(-lambda / r_prime * exp(1.0 / (r_prime - 1)))) :46 */
                vort[k] = (float)((double)zeta0 *
                    (1.0 - exp((double)(-lambda / r_prime) * exp(1.0 / (double)(r_prime - 1)))));
            } else {
                vort[k] = 0;
            }
        }
    }
}

/* field_generator.cpp:6-28 */
void fbo_add_cake_kuo2004(int nx, int ny, float lx, float ly, float *data,
                          float cx, float cy, float zeta_0, float scale_r)
{
    const float DX = lx / nx, DY = ly / ny;      /* configuration.hpp:23-24 */
    /* reference loops j<XPTS for y and i<YPTS for x (field_generator.cpp:14-20): square only */
    for (size_t j = 0; j < (size_t)nx; ++j) {
        float y = j * DY;
        for (size_t i = 0; i < (size_t)ny; ++i) {
            float x = i * DX;
            /* dist(): sqrtf(pow(x-cx,2.0) + pow(y-cy,2.0)) */
            float r = sqrtf((float)(pow((double)(x - cx), 2.0) + pow((double)(y - cy), 2.0))) / scale_r;
            if (r < 1) {
                /* data += zeta_0 * (1 - exp(-30.0 / r * exp(1.0/(r - 1.0)))) : double expr, += in float */
                data[(size_t)ny * i + j] = (float)((double)data[(size_t)ny * i + j] +
                    (double)zeta_0 * (1 - exp(-30.0 / (double)r * exp(1.0 / ((double)r - 1.0)))));
            }
        }
    }
}

/* makefield-Kuo2004.cpp:30-41; the reference forgets to zero its malloc */
void fbo_make_kuo2004(int nx, int ny, float lx, float ly, float *vort)
{
    memset(vort, 0, sizeof(float) * (size_t)nx * ny);
    fbo_add_cake_kuo2004(nx, ny, lx, ly, vort, lx / 2.0, ly / 2.0, 1.5e-2, 10000.0);
    fbo_add_cake_kuo2004(nx, ny, lx, ly, vort, lx / 2.0 + 50000.0, ly / 2.0, 3e-3, 30000.0);
}

/* makefield-gaussian.cpp:14-31 */
void fbo_make_gaussian(int nx, int ny, float lx, float ly, float *vort)
{
    float centerx = lx / 2.0, centery = ly / 2.0, zeta0 = 1e-3;
    float dx = lx / nx, dy = ly / ny;
    for (int i = 0; i < nx; ++i) {
        float x = i * dx;
        for (int j = 0; j < ny; ++j) {
            float y = j * dy;
            float r = sqrtf((float)(pow((double)(x - centerx), 2) + pow((double)(y - centery), 2)));
            vort[(size_t)ny * i + j] = (float)((double)zeta0 * exp(-pow((double)r / 60000.0, 2.0)));
        }
    }
}

/* makefield-const-vortex.cpp:14-38 */
void fbo_make_const_vortex(int nx, int ny, float lx, float ly, float *vort)
{
    float centerx = lx / 2.0, centery = ly / 2.0, r_bound = 6000.0, zeta0 = 2e-5;
    float dx = lx / nx, dy = ly / ny;
    for (int i = 0; i < nx; ++i) {
        float x = i * dx;
        for (int j = 0; j < ny; ++j) {
            float y = j * dy;
            float r = sqrtf((float)(pow((double)(x - centerx), 2) + pow((double)(y - centery), 2)));
            vort[(size_t)ny * i + j] = (r <= r_bound) ? zeta0 : 0;
        }
    }
}

/* ------------------------------------------------------------------------------------ */
/* field I/O: fieldio.cpp:7-33 (raw float32, host byte order, no header)                   */
/* ------------------------------------------------------------------------------------ */
int fbo_write_field(const char *filename, const float *data, size_t len)
{
    FILE *file = fopen(filename, "wb");
    if (!file) { perror("Write field."); return -1; }
    size_t n = fwrite(data, sizeof(float), len, file);
    fclose(file);
    fprintf(stderr, "Output %s\n", filename);                /* fieldio.cpp:18 */
    return n == len ? 0 : -2;
}

int fbo_read_field(const char *filename, float *data, size_t len)
{
    FILE *file = fopen(filename, "rb");
    if (!file) { perror("Read field."); return -1; }
    size_t n = fread(data, sizeof(float), len, file);
    fclose(file);
    fprintf(stderr, "%d bytes read: %s\n", (int)n, filename); /* fieldio.cpp:32 (elements, labelled bytes) */
    return n == len ? 0 : -2;
}

/* vorticity_source.cpp:112-133 */
int fbo_fifo_read(void *FILE_ptr, float *vort_src, size_t grids)
{
    FILE *fifo = (FILE *)FILE_ptr;
    char new_flag;
    if (fread(&new_flag, sizeof(char), 1, fifo) != 1) {
        fprintf(stderr, "No flag was detected, assume flag = 0\n"); fflush(stderr);
        return 1;
    }
    if (((unsigned int)new_flag) == 1) {
        if (fread(vort_src, sizeof(float), grids, fifo) != grids) {
            fprintf(stderr, "ERROR: Cannot read vorticity source input.\n"); fflush(stderr);
            return 2;
        }
        fprintf(stderr, "New vorticity source was given.\n");
    } else {
        fprintf(stderr, "No new vorticity source input was given.\n"); fflush(stderr);
    }
    return 0;
}
