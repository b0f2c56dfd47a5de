"""ctypes binding of include/fftbaro.h (the C-ABI drop-in boundary).

Host-side mirror of the reference's operator interface: `FftwfOperation` carries the method
names of `fftwf_operation<XPTS,YPTS>` (fftwfop.hpp:9-29) and `Model` the surface of the
main.cpp RK4 driver.  Arrays on the GPU are torch tensors (device memory + streams only);
the compute is entirely in libfftbaro.so.  There is no CPU fallback: a missing library or a
missing GPU raises.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

_lib = None
# fb_alltoall_fn (include/fftbaro.h): user, send, recv, stride, offset, count, hip stream
ALLTOALL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p)


class FftBaroError(RuntimeError):
    pass


def lib():
    """Loads libfftbaro.so (building it if the sources are newer); raises if unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own HIP runtime: it must be in the process before libfftbaro.so is
    # dlopen'ed, so that both resolve to ONE libamdhip64 (two runtimes cannot share a device).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    path = os.environ.get("FFTBARO_LIB") or _build.LIB      # developer hook: A/B a differently built library
    if path == _build.LIB and _build.stale():
        # atomic + locked (build.py): ranks of one job never see a half-written library.  A stale library that
        # cannot be rebuilt is an error -- tests must not run against old kernels (FFTBARO_ALLOW_STALE=1 overrides).
        try:
            _build.build_lib()
        except Exception as e:
            if not (os.environ.get("FFTBARO_ALLOW_STALE") and os.path.exists(path)):
                raise FftBaroError("libfftbaro.so is missing or stale and could not be rebuilt: %s" % e)
    L = C.CDLL(path)
    vp, fp, ip = C.c_void_p, C.c_void_p, C.c_int
    L.fb_strerror.restype = C.c_char_p
    L.fb_strerror.argtypes = [ip]
    L.fb_last_error.restype = C.c_char_p
    L.fb_version.restype = ip
    L.fb_size_supported.argtypes = [ip, ip]
    L.fb_device_count.argtypes = [C.POINTER(ip)]
    L.fb_set_device.argtypes = [ip]
    L.fb_create.argtypes = [C.POINTER(vp), ip, ip, C.c_float, C.c_float]
    L.fb_destroy.argtypes = [vp]
    L.fb_set_stream.argtypes = [vp, vp]
    L.fb_synchronize.argtypes = [vp]
    L.fb_get_tables.argtypes = [vp] + [C.c_void_p] * 5
    L.fb_malloc.argtypes = [C.POINTER(vp), C.c_size_t]
    L.fb_free.argtypes = [vp]
    L.fb_memcpy_h2d.argtypes = [vp, vp, vp, C.c_size_t]
    L.fb_memcpy_d2h.argtypes = [vp, vp, vp, C.c_size_t]
    L.fb_memset0.argtypes = [vp, vp, C.c_size_t]
    for n in ("fb_gradx", "fb_grady", "fb_laplacian", "fb_invert_laplacian", "fb_dealiase"):
        getattr(L, n).argtypes = [vp, fp, fp]
    L.fb_r2c.argtypes = [vp, fp, fp]
    L.fb_c2r.argtypes = [vp, fp, fp, ip]
    L.fb_backward_normalize.argtypes = [vp, fp]
    L.fb_negate.argtypes = [vp, fp]
    L.fb_jacobian.argtypes = [vp, fp, fp, fp, fp, fp, fp]
    L.fb_spec_axpy.argtypes = [vp, fp, fp, C.c_float]
    L.fb_spec_evolve.argtypes = [vp, fp, fp, C.c_float, fp]
    L.fb_spec_rk4_combine.argtypes = [vp, fp, fp, fp, fp, fp, C.c_float, fp]
    L.fb_model_create.argtypes = [C.POINTER(vp), vp, C.c_float, C.c_float]
    L.fb_model_destroy.argtypes = [vp]
    L.fb_model_set_vort.argtypes = [vp, fp]
    L.fb_model_set_source.argtypes = [vp, fp]
    L.fb_model_step.argtypes = [vp, ip]
    L.fb_model_use_graph.argtypes = [vp, ip]
    L.fb_model_get_vort.argtypes = [vp, fp]
    L.fb_model_get_diag.argtypes = [vp, fp, fp, fp]
    L.fb_model_get_spectrum.argtypes = [vp, fp]
    L.fb_model_set_spectrum.argtypes = [vp, fp]
    L.fb_model_info.argtypes = [vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.fb_model_time_steps.argtypes = [vp, ip, C.POINTER(C.c_float)]
    L.fb_model_profile_steps.argtypes = [vp, ip, C.POINTER(C.c_float), C.POINTER(C.c_int)]
    L.fb_make_field.argtypes = [C.c_char_p, ip, ip, C.c_float, C.c_float, C.c_void_p]
    L.fb_make_source_kuo2004.argtypes = [ip, ip, C.c_float, C.c_float, C.c_float, C.c_void_p]
    L.fb_create_slab.argtypes = [C.POINTER(vp), ip, ip, C.c_float, C.c_float, ip, ip]
    L.fb_slab_unique_id.argtypes = [C.c_char_p]
    L.fb_slab_create.argtypes = [C.POINTER(vp), ip, ip, C.c_float, C.c_float, C.c_float, C.c_float, ip, ip]
    L.fb_slab_destroy.argtypes = [vp]
    L.fb_slab_connect_rccl.argtypes = [vp, C.c_char_p]
    L.fb_local_hub_create.argtypes = [C.POINTER(vp), ip]
    L.fb_local_hub_destroy.argtypes = [vp]
    L.fb_slab_connect_local.argtypes = [vp, vp]
    L.fb_slab_connect_callback.argtypes = [vp, ALLTOALL_FN, vp]
    L.fb_slab_set_vort_local.argtypes = [vp, fp]
    L.fb_slab_set_source_local.argtypes = [vp, fp]
    L.fb_slab_get_vort_local.argtypes = [vp, fp]
    L.fb_slab_get_diag_local.argtypes = [vp, fp, fp, fp]
    L.fb_slab_step.argtypes = [vp, ip]
    L.fb_slab_synchronize.argtypes = [vp]
    L.fb_slab_time_steps.argtypes = [vp, ip, C.POINTER(C.c_float)]
    L.fb_slab_transport_selftest.argtypes = [vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.fb_slab_info.argtypes = [vp] + [C.POINTER(ip)] * 7
    L.fb_slab_transport_info.argtypes = [vp, C.c_char_p, C.c_size_t] + [C.POINTER(ip)] * 4
    L.fb_slab_geometry.argtypes = [ip, ip, ip] + [C.POINTER(ip)] * 3
    L.fb_slab_plan.argtypes = [ip, ip, ip, C.POINTER(ip), C.POINTER(ip), C.POINTER(ip), ip]
    L.fb_write_field.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t]
    L.fb_read_field.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t]
    _lib = L
    return L


EXPORTS = [
    "fb_strerror", "fb_last_error", "fb_version", "fb_size_supported", "fb_device_count", "fb_set_device", "fb_create", "fb_destroy", "fb_set_stream",
    "fb_synchronize", "fb_get_tables", "fb_malloc", "fb_free", "fb_memcpy_h2d", "fb_memcpy_d2h", "fb_memset0",
    "fb_gradx", "fb_grady", "fb_laplacian", "fb_invert_laplacian", "fb_dealiase", "fb_r2c", "fb_c2r",
    "fb_backward_normalize", "fb_negate", "fb_jacobian", "fb_spec_axpy", "fb_spec_evolve", "fb_spec_rk4_combine",
    "fb_model_create", "fb_model_destroy", "fb_model_set_vort", "fb_model_set_source", "fb_model_step",
    "fb_model_use_graph", "fb_model_get_vort", "fb_model_get_diag", "fb_model_get_spectrum", "fb_model_set_spectrum", "fb_model_info",
    "fb_model_time_steps", "fb_model_profile_steps", "fb_write_field", "fb_read_field", "fb_make_field", "fb_make_source_kuo2004",
    "fb_create_slab", "fb_slab_unique_id", "fb_slab_create", "fb_slab_destroy", "fb_slab_connect_rccl", "fb_local_hub_create",
    "fb_local_hub_destroy", "fb_slab_connect_local", "fb_slab_connect_callback", "fb_slab_set_vort_local", "fb_slab_set_source_local",
    "fb_slab_get_vort_local", "fb_slab_get_diag_local", "fb_slab_step", "fb_slab_synchronize", "fb_slab_time_steps", "fb_slab_transport_selftest", "fb_slab_transport_info", "fb_slab_info", "fb_slab_geometry", "fb_slab_plan", "fb_slab_col_groups",
    "fb_malloc_host", "fb_free_host", "fb_stream_create", "fb_stream_destroy", "fb_stream_synchronize", "fb_event_create", "fb_event_create_timing", "fb_event_elapsed_ms",
    "fb_event_destroy", "fb_event_record", "fb_stream_wait_event", "fb_event_synchronize", "fb_memcpy_d2h_async", "fb_memcpy_h2d_async", "fb_slab_record_event", "fb_slab_wait_event",
]


def check(status):
    if status != 0:
        L = lib()
        raise FftBaroError("%s: %s" % (L.fb_strerror(status).decode(), L.fb_last_error().decode()))


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise FftBaroError("no GPU visible: the engine has no CPU fallback")
    return torch


def _ptr(t):
    return C.c_void_p(t.data_ptr())


class FftwfOperation:
    """Mirror of `fftwf_operation<XPTS,YPTS>` (fftwfop.hpp:9-29) on device buffers.

    Spectra are torch complex64 tensors [nx, ny/2+1] on the GPU; `in is out` is allowed.
    """

    def __init__(self, nx, ny, Lx, Ly, stream=None):
        self.torch = _torch()
        self.nx, self.ny, self.hy = nx, ny, ny // 2 + 1
        h = C.c_void_p()
        check(lib().fb_create(C.byref(h), nx, ny, Lx, Ly))
        self._h = h
        self.use_current_stream()

    def use_current_stream(self):
        s = self.torch.cuda.current_stream().cuda_stream
        check(lib().fb_set_stream(self._h, C.c_void_p(s)))

    def close(self):
        if getattr(self, "_h", None):
            lib().fb_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers
    def empty_spec(self):
        return self.torch.empty((self.nx, self.hy), dtype=self.torch.complex64, device="cuda")

    def empty_real(self):
        return self.torch.empty((self.nx, self.ny), dtype=self.torch.float32, device="cuda")

    def _spec(self, t):
        assert t.is_cuda and t.dtype == self.torch.complex64 and t.is_contiguous() and tuple(t.shape) == (self.nx, self.hy)
        return _ptr(t)

    def _real(self, t):
        assert t.is_cuda and t.dtype == self.torch.float32 and t.is_contiguous() and tuple(t.shape) == (self.nx, self.ny)
        return _ptr(t)

    def _op(self, fn, a, out):
        out = self.empty_spec() if out is None else out
        check(fn(self._h, self._spec(a), self._spec(out)))
        return out

    # -- fftwfop.hpp:20-24
    def gradx(self, a, out=None): return self._op(lib().fb_gradx, a, out)
    def grady(self, a, out=None): return self._op(lib().fb_grady, a, out)
    def laplacian(self, a, out=None): return self._op(lib().fb_laplacian, a, out)
    def invertLaplacian(self, a, out=None): return self._op(lib().fb_invert_laplacian, a, out)
    def dealiase(self, a, out=None): return self._op(lib().fb_dealiase, a, out)

    # -- fftwfop.hpp:26-28
    def reflectedXWavenumberIndex(self, i):
        assert i >= 1
        return self.nx - i

    def HIDX(self, i, j): return self.hy * i + j
    def R_HIDX(self, i, j): return self.HIDX(self.reflectedXWavenumberIndex(i), j)

    # -- what the driver takes from FFTW (main.cpp:126-135,154,...)
    def r2c(self, real, out=None):
        out = self.empty_spec() if out is None else out
        check(lib().fb_r2c(self._h, self._real(real), self._spec(out)))
        return out

    def c2r(self, spec, out=None, normalize=False):
        out = self.empty_real() if out is None else out
        check(lib().fb_c2r(self._h, self._spec(spec), self._real(out), 1 if normalize else 0))
        return out

    # -- driver lambdas
    def backward_normalize(self, real): check(lib().fb_backward_normalize(self._h, self._real(real))); return real
    def negate(self, real): check(lib().fb_negate(self._h, self._real(real))); return real

    def jacobian(self, u, v, dzdx, dzdy, src=None, out=None):
        out = self.empty_real() if out is None else out
        check(lib().fb_jacobian(self._h, self._real(u), self._real(v), self._real(dzdx), self._real(dzdy),
                                self._real(src) if src is not None else None, self._real(out)))
        return out

    def spec_axpy(self, acc, x, a): check(lib().fb_spec_axpy(self._h, self._spec(acc), self._spec(x), a)); return acc

    def spec_evolve(self, base, rk, a, out=None):
        out = self.empty_spec() if out is None else out
        check(lib().fb_spec_evolve(self._h, self._spec(base), self._spec(rk), a, self._spec(out)))
        return out

    def spec_rk4_combine(self, base, k1, k2, k3, k4, dt, out=None):
        out = self.empty_spec() if out is None else out
        check(lib().fb_spec_rk4_combine(self._h, self._spec(base), self._spec(k1), self._spec(k2), self._spec(k3),
                                        self._spec(k4), dt, self._spec(out)))
        return out

    def tables(self):
        n, h = self.nx, self.hy
        gx = np.empty(n, np.float32); gy = np.empty(h, np.float32)
        lap = np.empty((n, h), np.float32); lapi = np.empty((n, h), np.float32); mask = np.empty((n, h), np.float32)
        check(lib().fb_get_tables(self._h, gx.ctypes.data, gy.ctypes.data, lap.ctypes.data, lapi.ctypes.data, mask.ctypes.data))
        return gx, gy, lap, lapi, mask

    def synchronize(self): check(lib().fb_synchronize(self._h))


class Model:
    """The main.cpp RK4 driver state (main.cpp:103-317) resident in HBM, fused stepping."""

    def __init__(self, nx, ny=None, Lx=600000.0, Ly=600000.0, nu=6.5, dt=3.0):
        ny = ny or nx
        self.fop = FftwfOperation(nx, ny, Lx, Ly)
        self.torch = self.fop.torch
        self.nx, self.ny, self.hy = nx, ny, ny // 2 + 1
        h = C.c_void_p()
        check(lib().fb_model_create(C.byref(h), self.fop._h, nu, dt))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib().fb_model_destroy(self._h)
            self._h = None
        self.fop.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _dev(self, a):
        t = self.torch
        if isinstance(a, np.ndarray):
            a = t.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
        assert a.is_cuda and a.dtype == t.float32 and a.is_contiguous() and tuple(a.shape) == (self.nx, self.ny)
        # the engine reads the buffer on ITS stream: whatever torch still has queued for it (a fill, a copy) must have landed
        t.cuda.current_stream().synchronize()
        return a

    def set_vort(self, vort): a = self._dev(vort); check(lib().fb_model_set_vort(self._h, _ptr(a))); self.fop.synchronize()

    def set_source(self, src):
        if src is None:
            check(lib().fb_model_set_source(self._h, None))
        else:
            a = self._dev(src); check(lib().fb_model_set_source(self._h, _ptr(a))); self.fop.synchronize()

    def step(self, n=1): check(lib().fb_model_step(self._h, n))

    def use_graph(self, enable=True):
        """hipGraph replay of the step; call under a non-default torch stream (after fop.use_current_stream())."""
        check(lib().fb_model_use_graph(self._h, 1 if enable else 0))

    def time_steps(self, n):
        ms = C.c_float()
        check(lib().fb_model_time_steps(self._h, n, C.byref(ms)))
        return ms.value

    KERNEL_CLASSES = ("k_col_strided_bwd4", "k_row_fused", "k_col_strided_fwd1", "k_col_mid")

    def profile_steps(self, n):
        """HIP-event time per kernel class over n steps: {class: (total_ms, launches)}."""
        ms = (C.c_float * 4)(); cnt = (C.c_int * 4)()
        check(lib().fb_model_profile_steps(self._h, n, ms, cnt))
        return {k: (ms[i], cnt[i]) for i, k in enumerate(self.KERNEL_CLASSES)}

    def vort(self):
        out = self.fop.empty_real(); check(lib().fb_model_get_vort(self._h, _ptr(out))); return out

    def diag(self):
        psi, u, v = self.fop.empty_real(), self.fop.empty_real(), self.fop.empty_real()
        check(lib().fb_model_get_diag(self._h, _ptr(psi), _ptr(u), _ptr(v)))
        return psi, u, v

    def spectrum(self):
        out = self.fop.empty_spec(); check(lib().fb_model_get_spectrum(self._h, _ptr(out))); return out

    def set_spectrum(self, spec): check(lib().fb_model_set_spectrum(self._h, self.fop._spec(spec)))

    def info(self):
        a, b = C.c_size_t(), C.c_size_t()
        check(lib().fb_model_info(self._h, C.byref(a), C.byref(b)))
        return {"hbm_bytes": a.value, "alg_bytes_per_step": b.value}


def write_field(path, data):
    data = np.ascontiguousarray(data, dtype=np.float32)
    check(lib().fb_write_field(path.encode(), data.ctypes.data, data.size))


def read_field(path, n):
    out = np.empty(n, dtype=np.float32)
    check(lib().fb_read_field(path.encode(), out.ctypes.data, n))
    return out


def make_field(kind, nx, ny=None, Lx=600000.0, Ly=600000.0):
    """Host-side initial vorticity (makefield-*.cpp restated with run-time grid size)."""
    ny = ny or nx
    out = np.empty((nx, ny), dtype=np.float32)
    check(lib().fb_make_field(kind.encode(), nx, ny, Lx, Ly, out.ctypes.data))
    return out


def make_source_kuo2004(nx, ny=None, Lx=600000.0, Ly=600000.0, duration=10800.0):
    ny = ny or nx
    out = np.empty((nx, ny), dtype=np.float32)
    check(lib().fb_make_source_kuo2004(nx, ny, Lx, Ly, duration, out.ctypes.data))
    return out
