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

  oracle_8192_step300.npz   : (round 3; retired in round 4: oracle_8192_step1000.npz from `oracle8192_1000` reproduces it bit for bit at steps 10 / 100 / 300
      and goes on to 600 and 1000)  the oracle on BASELINE configs[3] -- 8192^2 gaussian vortex (makefield-gaussian.cpp:14-31),
      dt = 0.375 s -- vort[::32, ::32] after 10, 100 and 300 steps, full-field L2 norms and sums (~1 h on 5 cores, 5 GiB).
  oracle_16384_src_step12.npz : the oracle on BASELINE configs[4] -- 16384^2 Kuo2004 initial field, the source-forced loop
      of main-shallow-water.cpp:277-338 with the FIFO producer's schedule (vort_src_input.cpp:35-61) shifted so that it is
      active: the cake 3e-3/duration at (L/2 + 50 km, L/2), R = 30 km arrives with flag 1 before step 2 and the zeroed field
      before step 6; dt = 0.1875 s -- vort[::64, ::64] after 1, 5, 8 and 12 steps, L2 norms and sums (~15 min on 8 cores,
      ~19 GiB).

None of them is a reference output: the reference's FFT-dependent programs cannot be built here (no FFTW) and the
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


def _snap(out, tag, v, sub):
    v64 = v.astype(np.float64)
    out["vort_sub%d_step%s" % (sub, tag)] = v[::sub, ::sub].copy()
    out["l2_step%s" % tag] = np.float64(np.sqrt((v64 * v64).sum()))
    out["sum_step%s" % tag] = np.float64(v64.sum())


def oracle8192():
    import oracle_py as O
    n, dt, sub = 8192, 0.375, 32
    m = O.Model(n, n, dt=dt)
    m.set_vort(O.make_field("gaussian", n))
    out = {"note": np.array("oracle/liboracle.so, 8192^2 gaussian (makefield-gaussian.cpp:14-31), nu=6.5, L=600 km, dt=0.375 s; "
                            "vort[::32, ::32], full-field l2 = sqrt(sum(vort^2)) in float64; made by "
                            "tests/golden/make_long_fixtures.py oracle8192")}
    done, t0 = 0, time.time()
    for upto in (10, 100, 300):
        while done < upto:
            m.step(10)
            done += 10
            print("step %d  %.0f s" % (done, time.time() - t0), flush=True)
        _snap(out, str(upto), m.vort(), sub)
        np.savez_compressed(os.path.join(HERE, "oracle_8192_step300.npz"), **out)
    print("done in %.0f s" % (time.time() - t0))


def oracle8192_1000():
    """configs[3] to the north-star horizon: steps 10 / 100 / 300 / 600 / 1000."""
    import oracle_py as O
    n, dt, sub = 8192, 0.375, 32
    m = O.Model(n, n, dt=dt)
    m.set_vort(O.make_field("gaussian", n))
    out = {"note": np.array("oracle/liboracle.so, 8192^2 gaussian (makefield-gaussian.cpp:14-31), nu=6.5, L=600 km, dt=0.375 s; "
                            "vort[::32, ::32], full-field l2 = sqrt(sum(vort^2)) in float64; made by "
                            "tests/golden/make_long_fixtures.py oracle8192_1000")}
    done, t0 = 0, time.time()
    for upto in (10, 100, 300, 600, 1000):
        while done < upto:
            m.step(10)
            done += 10
            print("step %d  %.0f s" % (done, time.time() - t0), flush=True)
        _snap(out, str(upto), m.vort(), sub)
        np.savez_compressed(os.path.join(HERE, "oracle_8192_step1000.npz"), **out)
    print("done in %.0f s" % (time.time() - t0))


def oracle16384_52():
    """configs[4] over 52 steps across the source's on and off: the cake arrives before step 2 and the field of zeros before
    step 27 (vort_src_input.cpp:40-55, shifted); records after steps 12, 26, 27, 40 and 52."""
    import oracle_py as O
    n, dt, sub, L = 16384, 0.1875, 64, 600000.0
    on_step, off_step = 2, 27
    m = O.Model(n, n, dt=dt)
    m.set_vort(O.make_field("kuo2004", n))
    out = {"note": np.array("oracle/liboracle.so, 16384^2 kuo2004 + source (main-shallow-water.cpp:277-338): cake 3e-3/10800 at "
                            "(L/2+50 km, L/2), R=30 km handed over before step 2, zeros before step 27; nu=6.5, L=600 km, "
                            "dt=0.1875 s; vort[::64, ::64]; made by tests/golden/make_long_fixtures.py oracle16384_52"),
           "on_step": np.int64(on_step), "off_step": np.int64(off_step)}
    t0 = time.time()
    for step in range(1, 53):
        if step == on_step:
            src = np.zeros((n, n), dtype=np.float32)
            O.add_cake(src, L, L, L / 2 + 50000.0, L / 2, 3e-3 / 10800.0, 30000.0)      # vort_src_input.cpp:46
            m.set_source(src)
            del src
        elif step == off_step:
            m.set_source(np.zeros((n, n), dtype=np.float32))                              # vort_src_input.cpp:52-55
        m.step(1)
        print("step %d  %.0f s" % (step, time.time() - t0), flush=True)
        if step in (12, 26, 27, 40, 52):
            _snap(out, str(step), m.vort(), sub)
            np.savez_compressed(os.path.join(HERE, "oracle_16384_src_step52.npz"), **out)
    print("done in %.0f s" % (time.time() - t0))


def _strip(nx, ny, dt, tag):
    """1000 steps on a strip-shaped grid: the cheap way to give the kernels that carry nx = 16384 (k_col_strided<128>, k_col_mid<128>) and
    ny = 16384 (k_rowh<2>) a horizon of 1000 steps (the 16384^2 grid itself costs the oracle 100 s of this container per step).  Elliptic
    vortex (makefield-elliptic-vortex.cpp:14-52, max wind ~ 83 m/s), dt = 0.375 s: U k_max dt ~ 1.8 at the largest retained wavenumber
    of the long direction (RK4 limit 2.8)."""
    import oracle_py as O
    m = O.Model(nx, ny, dt=dt)
    m.set_vort(O.make_field("elliptic", nx, ny))
    sx, sy = max(1, nx // 256), max(1, ny // 256)
    out = {"note": np.array("oracle/liboracle.so, %d x %d elliptic vortex, nu=6.5, L=600 km, dt=%g s; vort[::%d, ::%d], l2 = sqrt(sum(vort^2)) in "
                            "float64; made by tests/golden/make_long_fixtures.py %s" % (nx, ny, dt, sx, sy, tag)),
           "sub": np.array([sx, sy], dtype=np.int64), "dt": np.float64(dt)}
    done, t0 = 0, time.time()
    for upto in (100, 500, 1000):
        while done < upto:
            m.step(50)
            done += 50
            print("step %d  %.0f s" % (done, time.time() - t0), flush=True)
        v = m.vort()
        v64 = v.astype(np.float64)
        out["vort_step%d" % upto] = v[::sx, ::sy].copy()
        out["l2_step%d" % upto] = np.float64(np.sqrt((v64 * v64).sum()))
        out["sum_step%d" % upto] = np.float64(v64.sum())
        np.savez_compressed(os.path.join(HERE, "oracle_%dx%d_step1000.npz" % (nx, ny)), **out)
    print("done in %.0f s" % (time.time() - t0))


def strip_cols():
    _strip(16384, 64, 0.375, "strip_cols")


def strip_rows():
    _strip(128, 16384, 0.375, "strip_rows")


def oracle16384():
    import oracle_py as O
    n, dt, sub, L = 16384, 0.1875, 64, 600000.0
    on_step, off_step = 2, 6                       # the producer's beg_step / end_step (vort_src_input.cpp:40-41), shifted
    m = O.Model(n, n, dt=dt)
    m.set_vort(O.make_field("kuo2004", n))
    out = {"note": np.array("oracle/liboracle.so, 16384^2 kuo2004 + source (main-shallow-water.cpp:277-338): cake 3e-3/10800 at "
                            "(L/2+50 km, L/2), R=30 km handed over before step 2, zeros before step 6; nu=6.5, L=600 km, "
                            "dt=0.1875 s; vort[::64, ::64]; made by tests/golden/make_long_fixtures.py oracle16384"),
           "on_step": np.int64(on_step), "off_step": np.int64(off_step)}
    t0 = time.time()
    for step in range(1, 13):
        if step == on_step:
            src = np.zeros((n, n), dtype=np.float32)
            O.add_cake(src, L, L, L / 2 + 50000.0, L / 2, 3e-3 / 10800.0, 30000.0)      # vort_src_input.cpp:46
            m.set_source(src)
            del src
        elif step == off_step:
            m.set_source(np.zeros((n, n), dtype=np.float32))                              # vort_src_input.cpp:52-55
        m.step(1)
        print("step %d  %.0f s" % (step, time.time() - t0), flush=True)
        if step in (1, 5, 8, 12):
            _snap(out, str(step), m.vort(), sub)
            np.savez_compressed(os.path.join(HERE, "oracle_16384_src_step12.npz"), **out)
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
    {"oracle4096": oracle4096, "fp64_1024": fp64_1024, "oracle8192": oracle8192, "oracle16384": oracle16384,
     "oracle8192_1000": oracle8192_1000, "oracle16384_52": oracle16384_52, "strip_cols": strip_cols, "strip_rows": strip_rows}[sys.argv[1]]()
