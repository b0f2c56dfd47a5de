"""Builds the native pieces in-tree (gfx950): lib/libfftbaro.so (HIP kernels + C ABI), lib/libfieldio.so (the
reference's fieldio symbols, host only) and lib/libfftw3f_fb.so (FFTW3-named entry points over the C ABI).

Builds are atomic (compile to a temporary file in lib/, then os.replace) and serialised with an fcntl lock, so that
the ranks of a torch.distributed.run job cannot dlopen a half-written library or compile on top of each other.
"""
import fcntl
import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIBDIR = os.path.join(HERE, "lib")
SRC = os.path.join(CSRC, "fftbaro.hip")
SRC_HOST = [os.path.join(CSRC, "fb_fields.cpp"), os.path.join(CSRC, "fb_fieldio.cpp"), os.path.join(CSRC, "fb_slab_comm.cpp")]
LIB = os.path.join(LIBDIR, "libfftbaro.so")
LIB_FIELDIO = os.path.join(LIBDIR, "libfieldio.so")
LIB_FFTW = os.path.join(LIBDIR, "libfftw3f_fb.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CXX = os.environ.get("CXX", "g++")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=on",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wno-unused-value"]
LINK = ["-ldl"]          # RCCL (the multi-GPU transposes) is dlopen'ed on first use, not linked


def _deps(lib):
    if lib == LIB:
        return [SRC] + SRC_HOST + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(INCLUDE, "*.h"))
    if lib == LIB_FIELDIO:
        return [os.path.join(CSRC, "fb_fieldio.cpp"), os.path.join(INCLUDE, "fftbaro.h")]
    return [os.path.join(HERE, "host", "fftw3f_fb.cpp"), os.path.join(INCLUDE, "fftw3_fb.h"), os.path.join(INCLUDE, "fftbaro.h")]


def _stale(lib):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(d) > t for d in _deps(lib))


def stale():
    return _stale(LIB) or _stale(LIB_FIELDIO) or _stale(LIB_FFTW)


def _cmd(lib, out):
    if lib == LIB:
        return [HIPCC] + FLAGS + ["-o", out, SRC] + SRC_HOST + LINK
    if lib == LIB_FIELDIO:
        return [CXX, "-std=c++11", "-O2", "-fPIC", "-shared", "-o", out, os.path.join(CSRC, "fb_fieldio.cpp")]
    return [CXX, "-std=c++11", "-O2", "-fPIC", "-shared", "-I" + INCLUDE, "-o", out, os.path.join(HERE, "host", "fftw3f_fb.cpp"),
            "-L" + LIBDIR, "-lfftbaro", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib"]


def build_lib(force=False, verbose=False):
    """Builds whatever is stale (everything with force=True); returns the path of libfftbaro.so."""
    os.makedirs(LIBDIR, exist_ok=True)
    with open(os.path.join(LIBDIR, ".build.lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)                       # one builder at a time; the others re-check staleness
        try:
            for lib in (LIB, LIB_FIELDIO, LIB_FFTW):
                if not (force or _stale(lib)):
                    continue
                tmp = "%s.tmp.%d" % (lib, os.getpid())
                cmd = _cmd(lib, tmp)
                if verbose:
                    print(" ".join(cmd))
                try:
                    subprocess.check_call(cmd)
                    os.replace(tmp, lib)
                finally:
                    if os.path.exists(tmp):
                        os.remove(tmp)
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)
    return LIB


if __name__ == "__main__":
    build_lib(force=True, verbose=True)
