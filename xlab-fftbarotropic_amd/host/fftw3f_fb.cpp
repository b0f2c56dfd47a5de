// fftw3f_fb.cpp -- FFTW3-named entry points (include/fftw3_fb.h) over the C ABI of the engine.
// One engine context per (n0, n1), created by the first plan of that size; Lx = Ly = 1 (the transforms do not
// depend on the domain size -- only the operator tables do, and those belong to fftwf_operation<>).
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>

#include "fftbaro.h"
#include "fftw3_fb.h"

struct fftwf_plan_s { fb_ctx *ctx; int c2r; float *real; float *spec; };

namespace {
std::mutex g_mu;
std::map<std::pair<int, int>, fb_ctx *> g_ctx;

fb_ctx *ctx_for(int n0, int n1)
{
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_ctx.find({n0, n1});
    if (it != g_ctx.end()) return it->second;
    fb_ctx *c = nullptr;
    if (fb_create(&c, n0, n1, 1.0f, 1.0f) != FB_OK) {
        std::fprintf(stderr, "fftwf_plan_dft_*_2d(%d, %d): %s\n", n0, n1, fb_last_error());
        return nullptr;                                   // FFTW's planners return NULL on failure
    }
    g_ctx[{n0, n1}] = c;
    return c;
}
void must(int status, const char *what)
{
    if (status != FB_OK) { std::fprintf(stderr, "%s: %s (%s)\n", what, fb_strerror(status), fb_last_error()); std::abort(); }   // fftwf_execute is void
}
}  // namespace

extern "C" {

void *fftwf_malloc(size_t n) { void *p = nullptr; return fb_malloc_host(&p, n) == FB_OK ? p : nullptr; }
void fftwf_free(void *p) { if (p && fb_free_host(p) != FB_OK) (void)fb_free(p); }     // also takes fb_malloc'ed device buffers
float *fftwf_alloc_real(size_t n) { return (float *)fftwf_malloc(n * sizeof(float)); }
fftwf_complex *fftwf_alloc_complex(size_t n) { return (fftwf_complex *)fftwf_malloc(n * sizeof(fftwf_complex)); }

fftwf_plan fftwf_plan_dft_r2c_2d(int n0, int n1, float *in, fftwf_complex *out, unsigned)
{
    fb_ctx *c = (in && out) ? ctx_for(n0, n1) : nullptr;
    return c ? new fftwf_plan_s{c, 0, in, (float *)out} : nullptr;
}
fftwf_plan fftwf_plan_dft_c2r_2d(int n0, int n1, fftwf_complex *in, float *out, unsigned)
{
    fb_ctx *c = (in && out) ? ctx_for(n0, n1) : nullptr;
    return c ? new fftwf_plan_s{c, 1, out, (float *)in} : nullptr;
}
void fftwf_execute(const fftwf_plan p)
{
    if (!p) { std::fprintf(stderr, "fftwf_execute: NULL plan\n"); std::abort(); }
    if (p->c2r) must(fb_c2r(p->ctx, p->spec, p->real, 0), "fftwf_execute(c2r)");
    else must(fb_r2c(p->ctx, p->real, p->spec), "fftwf_execute(r2c)");
    must(fb_synchronize(p->ctx), "fftwf_execute");       // FFTW is synchronous: the host reads the result next
}
void fftwf_destroy_plan(fftwf_plan p) { delete p; }
void fftwf_cleanup(void)
{
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto &kv : g_ctx) fb_destroy(kv.second);
    g_ctx.clear();
}

}  // extern "C"
