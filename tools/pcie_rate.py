#!/usr/bin/env python3
"""PCIe-inclusive rates of the host-buffer surfaces; run on the GPU box:
  * the FFTW-named shim on fftwf_malloc'ed (pinned host) buffers: one r2c + one c2r of n x n, operands over PCIe each call;
  * fb_model_set_vort / fb_model_get_vort from / to pageable numpy arrays (the driver's record path moves the same bytes).
usage: tools/pcie_rate.py [n=4096]"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import xlab_fftbarotropic_amd as X  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
X.lib()
F = C.CDLL(os.path.join(ROOT, "xlab-fftbarotropic_amd", "lib", "libfftw3f_fb.so"))
F.fftwf_malloc.restype = C.c_void_p
F.fftwf_malloc.argtypes = [C.c_size_t]
F.fftwf_plan_dft_r2c_2d.restype = C.c_void_p
F.fftwf_plan_dft_r2c_2d.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint]
F.fftwf_plan_dft_c2r_2d.restype = C.c_void_p
F.fftwf_plan_dft_c2r_2d.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint]
F.fftwf_execute.argtypes = [C.c_void_p]
F.fftwf_free.argtypes = [C.c_void_p]
h = n // 2 + 1
real = F.fftwf_malloc(4 * n * n)
spec = F.fftwf_malloc(8 * n * h)
a = np.ctypeslib.as_array(C.cast(real, C.POINTER(C.c_float)), shape=(n, n))
a[:] = np.random.default_rng(1).standard_normal((n, n)).astype(np.float32)
pf = F.fftwf_plan_dft_r2c_2d(n, n, real, spec, 1 << 6)
pb = F.fftwf_plan_dft_c2r_2d(n, n, spec, real, 1 << 6)
assert pf and pb
for _ in range(2):
    F.fftwf_execute(pf)
    F.fftwf_execute(pb)
    a *= 1.0 / (n * n)
reps = 5
t0 = time.perf_counter()
for _ in range(reps):
    F.fftwf_execute(pf)
t1 = time.perf_counter()
for _ in range(reps):
    F.fftwf_execute(pb)
t2 = time.perf_counter()
r2c_ms, c2r_ms = (t1 - t0) / reps * 1e3, (t2 - t1) / reps * 1e3
out = {"grid": n, "shim_r2c_ms": r2c_ms, "shim_c2r_ms": c2r_ms,
       "shim_bytes_per_transform": 4 * n * n + 8 * n * h,
       "shim_GBs": (4 * n * n + 8 * n * h) / ((r2c_ms + c2r_ms) / 2 * 1e-3) / 1e9,
       # main.cpp's RK4 step = 16 c2r + 4 r2c (+ host loops, not counted)
       "shim_steps_per_s_fft_only": 1e3 / (16 * c2r_ms + 4 * r2c_ms)}
m = X.Model(n, n)
v = X.make_field("kuo2004", n)
m.set_vort(v)
m.step(1)
m.fop.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    m.set_vort(v)
m.fop.synchronize()
t1 = time.perf_counter()
for _ in range(reps):
    g = m.vort().cpu().numpy()
t2 = time.perf_counter()
out["set_vort_ms"] = (t1 - t0) / reps * 1e3
out["get_vort_ms"] = (t2 - t1) / reps * 1e3
k = 50
m.step(40)                      # the device needs ~25 ms of this load to reach full speed
m.fop.synchronize()
t0 = time.perf_counter()
m.step(k)
m.fop.synchronize()
t1 = time.perf_counter()
out["step_ms"] = (t1 - t0) / k * 1e3
out["steps_per_s_resident"] = 1e3 / out["step_ms"]
out["steps_per_s_with_host_round_trip_every_step"] = 1e3 / (out["step_ms"] + out["set_vort_ms"] + out["get_vort_ms"])
print(json.dumps(out))
