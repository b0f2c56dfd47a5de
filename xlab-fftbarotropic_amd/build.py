"""Builds the HIP shared library (gfx950) in-tree: xlab-fftbarotropic_amd/lib/libfftbaro.so."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "fftbaro.hip")
SRC_HOST = [os.path.join(HERE, "csrc", "fb_fields.cpp")]
DEPS = [SRC] + SRC_HOST + [ os.path.join(HERE, "csrc", "fb_kernels.h"), os.path.join(HERE, "csrc", "fb_fft_core.h"), os.path.join(HERE, "csrc", "fb_col_full.h"), os.path.join(HERE, "csrc", "fb_row3.h"), os.path.join(HERE, "csrc", "fb_row8.h"),
        os.path.join(os.path.dirname(HERE), "include", "fftbaro.h")]
LIB = os.path.join(HERE, "lib", "libfftbaro.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wno-unused-value"]


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build_lib(force=False, verbose=False):
    if not (force or stale()):
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    cmd = [HIPCC] + FLAGS + ["-o", LIB, SRC] + SRC_HOST
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build_lib(force=True, verbose=True)
