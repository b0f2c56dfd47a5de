"""ctypes binding of the CPU ORACLE (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never from the product package.  See oracle/fb_oracle.h for the parity
pin status and the reference file:line each function restates.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")


def build(force=False):
    """Compile liboracle.so (and oracle/_ref when /root/reference exists)."""
    src = os.path.join(_HERE, "fb_oracle.c")
    stale = (not os.path.exists(_LIB)) or os.path.getmtime(_LIB) < os.path.getmtime(src)
    if force or stale or (os.path.isdir("/root/reference/src") and not os.path.isdir(os.path.join(_HERE, "_ref"))):
        subprocess.check_call(["make", "-s", "-C", _HERE], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        fp = C.POINTER(C.c_float)
        vp = C.c_void_p
        L.fbo_op_create.restype = vp
        L.fbo_op_create.argtypes = [C.c_int, C.c_int, C.c_float, C.c_float]
        L.fbo_op_destroy.argtypes = [vp]
        for name in ("fbo_gradx", "fbo_grady", "fbo_laplacian", "fbo_invert_laplacian", "fbo_dealiase"):
            getattr(L, name).argtypes = [vp, fp, fp]
            getattr(L, name).restype = None
        L.fbo_r2c_2d.argtypes = [C.c_int, C.c_int, fp, fp]
        L.fbo_c2r_2d.argtypes = [C.c_int, C.c_int, fp, fp]
        L.fbo_fft1d.argtypes = [C.c_int, C.c_int, fp]
        L.fbo_model_create.restype = vp
        L.fbo_model_create.argtypes = [C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float]
        L.fbo_model_destroy.argtypes = [vp]
        L.fbo_model_set_vort.argtypes = [vp, fp]
        L.fbo_model_set_source.argtypes = [vp, fp]
        L.fbo_model_step.argtypes = [vp]
        L.fbo_model_get_vort.argtypes = [vp, fp]
        L.fbo_model_get_diag.argtypes = [vp, fp, fp, fp]
        L.fbo_model_get_spectrum.argtypes = [vp, fp]
        L.fbo_model_set_spectrum.argtypes = [vp, fp]
        for name in ("fbo_make_elliptic", "fbo_make_kuo2004", "fbo_make_gaussian", "fbo_make_const_vortex"):
            getattr(L, name).argtypes = [C.c_int, C.c_int, C.c_float, C.c_float, fp]
            getattr(L, name).restype = None
        L.fbo_add_cake_kuo2004.argtypes = [C.c_int, C.c_int, C.c_float, C.c_float, fp,
                                           C.c_float, C.c_float, C.c_float, C.c_float]
        L.fbo_write_field.argtypes = [C.c_char_p, fp, C.c_size_t]
        L.fbo_read_field.argtypes = [C.c_char_p, fp, C.c_size_t]
        _lib = L
    return _lib


def _p(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_float))


class Operators:
    """fftwf_operation<XPTS,YPTS> restated (fftwfop.cpp:5-124). Spectra are complex64 [nx, ny/2+1]."""

    def __init__(self, nx, ny, lx, ly):
        self.nx, self.ny, self.hy = nx, ny, ny // 2 + 1
        self._h = lib().fbo_op_create(nx, ny, lx, ly)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().fbo_op_destroy(self._h)
            self._h = None

    def _apply(self, fn, spec):
        spec = np.ascontiguousarray(spec, dtype=np.complex64)
        assert spec.shape == (self.nx, self.hy)
        out = np.empty_like(spec)
        fn(self._h, _p(spec.view(np.float32)), _p(out.view(np.float32)))
        return out

    def gradx(self, s): return self._apply(lib().fbo_gradx, s)
    def grady(self, s): return self._apply(lib().fbo_grady, s)
    def laplacian(self, s): return self._apply(lib().fbo_laplacian, s)
    def invertLaplacian(self, s): return self._apply(lib().fbo_invert_laplacian, s)
    def dealiase(self, s): return self._apply(lib().fbo_dealiase, s)

    def tables(self):
        """(gradx_coe[nx], grady_coe[hy], lap[nx,hy], lap_inv[nx,hy], mask[nx,hy]) as float32 copies."""
        class _Op(C.Structure):
            _fields_ = [("nx", C.c_int), ("ny", C.c_int), ("hy", C.c_int), ("lx", C.c_float), ("ly", C.c_float),
                        ("gx", C.POINTER(C.c_float)), ("gy", C.POINTER(C.c_float)), ("lap", C.POINTER(C.c_float)),
                        ("lapi", C.POINTER(C.c_float)), ("mask", C.POINTER(C.c_float))]
        o = C.cast(self._h, C.POINTER(_Op)).contents
        n, h = self.nx, self.hy
        g = lambda p, cnt: np.ctypeslib.as_array(p, shape=(cnt,)).copy()
        return (g(o.gx, n), g(o.gy, h), g(o.lap, n * h).reshape(n, h), g(o.lapi, n * h).reshape(n, h),
                g(o.mask, n * h).reshape(n, h))


def r2c(field):
    field = np.ascontiguousarray(field, dtype=np.float32)
    nx, ny = field.shape
    out = np.empty((nx, ny // 2 + 1), dtype=np.complex64)
    lib().fbo_r2c_2d(nx, ny, _p(field), _p(out.view(np.float32)))
    return out


def c2r(spec, ny):
    spec = np.ascontiguousarray(spec, dtype=np.complex64)
    nx = spec.shape[0]
    assert spec.shape[1] == ny // 2 + 1
    out = np.empty((nx, ny), dtype=np.float32)
    lib().fbo_c2r_2d(nx, ny, _p(spec.view(np.float32)), _p(out))
    return out


def fft1d(x, sign):
    x = np.ascontiguousarray(x, dtype=np.complex64).copy()
    lib().fbo_fft1d(x.shape[0], sign, _p(x.view(np.float32)))
    return x


class Model:
    """RK4 driver state restated from main.cpp:103-317."""

    def __init__(self, nx, ny, lx=600000.0, ly=600000.0, nu=6.5, dt=3.0):
        self.nx, self.ny, self.hy = nx, ny, ny // 2 + 1
        self._h = lib().fbo_model_create(nx, ny, lx, ly, nu, dt)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().fbo_model_destroy(self._h)
            self._h = None

    def set_vort(self, vort):
        vort = np.ascontiguousarray(vort, dtype=np.float32)
        assert vort.shape == (self.nx, self.ny)
        lib().fbo_model_set_vort(self._h, _p(vort))

    def set_source(self, src):
        if src is None:
            lib().fbo_model_set_source(self._h, None)
        else:
            src = np.ascontiguousarray(src, dtype=np.float32)
            lib().fbo_model_set_source(self._h, _p(src))

    def step(self, n=1):
        for _ in range(n):
            lib().fbo_model_step(self._h)

    def vort(self):
        out = np.empty((self.nx, self.ny), dtype=np.float32)
        lib().fbo_model_get_vort(self._h, _p(out))
        return out

    def diag(self):
        psi = np.empty((self.nx, self.ny), dtype=np.float32)
        u = np.empty_like(psi)
        v = np.empty_like(psi)
        lib().fbo_model_get_diag(self._h, _p(psi), _p(u), _p(v))
        return psi, u, v

    def spectrum(self):
        out = np.empty((self.nx, self.hy), dtype=np.complex64)
        lib().fbo_model_get_spectrum(self._h, _p(out.view(np.float32)))
        return out

    def set_spectrum(self, spec):
        spec = np.ascontiguousarray(spec, dtype=np.complex64)
        lib().fbo_model_set_spectrum(self._h, _p(spec.view(np.float32)))


def make_field(kind, nx, ny=None, lx=600000.0, ly=600000.0):
    ny = ny or nx
    out = np.zeros((nx, ny), dtype=np.float32)
    fn = {"elliptic": lib().fbo_make_elliptic, "kuo2004": lib().fbo_make_kuo2004,
          "gaussian": lib().fbo_make_gaussian, "const": lib().fbo_make_const_vortex}[kind]
    fn(nx, ny, lx, ly, _p(out))
    return out


def add_cake(data, lx, ly, cx, cy, zeta0, scale_r):
    nx, ny = data.shape
    lib().fbo_add_cake_kuo2004(nx, ny, lx, ly, _p(data), cx, cy, zeta0, scale_r)


def write_field(path, data):
    data = np.ascontiguousarray(data, dtype=np.float32)
    return lib().fbo_write_field(path.encode(), _p(data.reshape(-1)), data.size)


def read_field(path, n):
    out = np.empty(n, dtype=np.float32)
    rc = lib().fbo_read_field(path.encode(), _p(out), n)
    return rc, out
