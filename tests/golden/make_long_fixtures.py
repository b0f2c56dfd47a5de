#!/usr/bin/env python3
"""Long-run fixtures for the north-star tolerance (<= 1e-5 relative L2 after 1000 RK4 steps, main.cpp:259-317).
Run in the build container only (the GPU box never regenerates them):

    OMP_NUM_THREADS=6 python tests/golden/make_long_fixtures.py oracle4096      # ~1 h on 6 cores
    python tests/golden/make_long_fixtures.py fp64_1024                           # ~15 min, one core

  oracle_4096_step1000.npz : oracle/liboracle.so (the C restatement of main.cpp:146-317, fp32, OpenMP build) on
      BASELINE configs[2] -- 4096^2 Kuo2004, dt = 0.75 s -- sub-sampled (every 16th point in x and y) after
      100, 500 and 1000 steps, plus the full-field L2 norms and the field sums at those steps.
  fp64_1024_step1000.npz   : tests/ref_numpy.py Model64 (independent fp64 numpy restatement on rfft2/irfft2) on
      configs[1] -- 1024^2 elliptic vortex, dt = 3 s -- sub-sampled (every 4th point) after 1000 steps: the
      oracle's own 1000-step cross-check.

Neither is a reference output: the reference's FFT-dependent programs cannot be built here (no FFTW) and the
reference holds no fixtures ("parity unpinned" by the reference).
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]


def oracle4096():
    import oracle_py as O
    n, dt, sub = 4096, 0.75, 16
    m = O.Model(n, n, dt=dt)
    m.set_vort(O.make_field("kuo2004", n))
    out = {"note": np.array("oracle/liboracle.so, 4096^2 kuo2004, nu=6.5, L=600 km, dt=0.75 s; vort[::16, ::16], "
                            "full-field l2 = sqrt(sum(vort^2)) in float64; made by tests/golden/make_long_fixtures.py")}
    done, t0 = 0, time.time()
    for upto in (100, 500, 1000):
        while done < upto:
            m.step(10)
            done += 10
            if done % 50 == 0:
                print("step %d  %.0f s" % (done, time.time() - t0), flush=True)
        v = m.vort()
        v64 = v.astype(np.float64)
        out["vort_sub16_step%d" % upto] = v[::sub, ::sub].copy()
        out["l2_step%d" % upto] = np.float64(np.sqrt((v64 * v64).sum()))
        out["sum_step%d" % upto] = np.float64(v64.sum())
        np.savez_compressed(os.path.join(HERE, "oracle_4096_step1000.npz"), **out)     # partial results survive a kill
    print("done in %.0f s" % (time.time() - t0))


def fp64_1024():
    import oracle_py as O
    import ref_numpy as R
    n, dt, sub = 1024, 3.0, 4
    v0 = O.make_field("elliptic", n)
    m = R.Model64(n, n, 600000.0, 600000.0, 6.5, dt)
    m.set_vort(v0)
    t0 = time.time()
    for s in range(10):
        m.step(100)
        print("step %d  %.0f s" % (100 * (s + 1), time.time() - t0), flush=True)
    v = m.vort()
    np.savez_compressed(os.path.join(HERE, "fp64_1024_step1000.npz"), vort_sub4=v[::sub, ::sub].astype(np.float64),
                        l2=np.float64(np.sqrt((v.astype(np.float64) ** 2).sum())),
                        note=np.array("tests/ref_numpy.py Model64 (fp64, numpy rfft2/irfft2), 1024^2 elliptic, dt=3 s, "
                                      "1000 steps; vort[::4, ::4]"))
    print("done in %.0f s" % (time.time() - t0))


if __name__ == "__main__":
    {"oracle4096": oracle4096, "fp64_1024": fp64_1024}[sys.argv[1]]()
