"""CPU tests of the ORACLE itself (runs with -m "not gpu").

Pins, in order of strength:
  1. reference's own fieldio / generators / FIFO producer (oracle/_ref, built from the
     reference sources unmodified; committed hashes in tests/golden/ref_meta.json);
  2. mathematical definition of the DFT (numpy fp64) and analytic known answers for the
     operators -- the FFT-dependent part of the reference is unbuildable here (no FFTW), and
     the reference ships no fixtures: "parity unpinned" by the reference for that part;
  3. committed fp64 fixtures (tests/golden/golden.npz, made by tests/golden/make_golden.py).
"""
import ctypes
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

import oracle_py as O
import ref_numpy as R

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "golden.npz"))
META = json.load(open(os.path.join(HERE, "golden", "ref_meta.json")))
REFDIR = os.path.join(os.path.dirname(HERE), "oracle", "_ref")
L = 600000.0


# ---------------------------------------------------------------- generators / fieldio
@pytest.mark.parametrize("kind", ["elliptic", "kuo2004", "gaussian", "const"])
def test_generators_match_reference_hash(kind):
    """Oracle generators reproduce the reference-built generators bit for bit at NPTS=768."""
    f = O.make_field(kind, 768)
    assert hashlib.sha256(f.tobytes()).hexdigest() == META[kind]["sha256"]
    assert np.array_equal(f[::16, ::16], GOLD["ref_gen_%s_sub16" % kind])


@pytest.mark.skipif(not os.path.exists(os.path.join(REFDIR, "libfieldio.so")), reason="oracle/_ref not built")
def test_fieldio_bytes_match_reference_library(tmp_path):
    """writeField/readField of the reference's own fieldio.cpp (fieldio.cpp:7-33) vs oracle."""
    ref = ctypes.CDLL(os.path.join(REFDIR, "libfieldio.so"))
    wr = getattr(ref, "_Z10writeFieldPKcPfm")
    rd = getattr(ref, "_Z9readFieldPKcPfm")
    for fn in (wr, rd):
        fn.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_float), ctypes.c_size_t]
        fn.restype = None
    a = np.random.default_rng(1).standard_normal(1000).astype(np.float32)
    p_ref, p_mine = str(tmp_path / "ref.bin"), str(tmp_path / "mine.bin")
    wr(p_ref.encode(), a.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), a.size)
    assert O.write_field(p_mine, a) == 0
    assert open(p_ref, "rb").read() == open(p_mine, "rb").read() == a.tobytes()
    b = np.zeros_like(a)
    rd(p_mine.encode(), b.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), b.size)
    rc, c = O.read_field(p_ref, a.size)
    assert rc == 0 and np.array_equal(b, a) and np.array_equal(c, a)


def test_fieldio_error_paths(tmp_path):
    rc, _ = O.read_field(str(tmp_path / "missing.bin"), 4)
    assert rc == -1
    p = str(tmp_path / "short.bin")
    np.arange(3, dtype=np.float32).tofile(p)
    rc, _ = O.read_field(p, 4)
    assert rc == -2                      # short read is reported, unlike fieldio.cpp:26


@pytest.mark.skipif(not os.path.exists(os.path.join(REFDIR, "vort_src_input.out")), reason="oracle/_ref not built")
def test_fifo_protocol_against_reference_producer(tmp_path):
    """The reference producer's byte stream (vort_src_input.cpp:43-61) is consumed flag by flag."""
    raw = subprocess.run([os.path.join(REFDIR, "vort_src_input.out")], stdout=subprocess.PIPE,
                         stderr=subprocess.DEVNULL, check=True).stdout
    assert hashlib.sha256(raw).hexdigest() == META["fifo_stream"]["sha256"]
    p = tmp_path / "stream.bin"
    p.write_bytes(raw)
    libc = ctypes.CDLL(None)
    libc.fopen.restype = ctypes.c_void_p
    libc.fopen.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
    libc.fclose.argtypes = [ctypes.c_void_p]
    fh = libc.fopen(str(p).encode(), b"rb")
    lib = O.lib()
    lib.fbo_fifo_read.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float), ctypes.c_size_t]
    src = np.zeros(768 * 768, dtype=np.float32)
    rcs = [lib.fbo_fifo_read(fh, src.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), src.size)
           for _ in range(1200)]       # consumer reads total_steps flags, producer wrote total_steps-1
    libc.fclose(fh)
    assert rcs[:1199] == [0] * 1199 and rcs[1199] == 1      # last read hits EOF


def test_fifo_protocol_new_field(tmp_path):
    n = 16
    field = np.arange(n, dtype=np.float32)
    p = tmp_path / "s.bin"
    p.write_bytes(b"\x00" + b"\x01" + field.tobytes() + b"\x01" + field.tobytes()[:8])
    libc = ctypes.CDLL(None)
    libc.fopen.restype = ctypes.c_void_p
    libc.fopen.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
    libc.fclose.argtypes = [ctypes.c_void_p]
    fh = libc.fopen(str(p).encode(), b"rb")
    lib = O.lib()
    lib.fbo_fifo_read.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float), ctypes.c_size_t]
    src = np.zeros(n, dtype=np.float32)
    ptr = src.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
    assert lib.fbo_fifo_read(fh, ptr, n) == 0 and not src.any()
    assert lib.fbo_fifo_read(fh, ptr, n) == 0 and np.array_equal(src, field)
    assert lib.fbo_fifo_read(fh, ptr, n) == 2          # short field
    libc.fclose(fh)


# ---------------------------------------------------------------- operator tables
@pytest.mark.parametrize("nx,ny", [(64, 64), (256, 256), (96, 64), (768, 768)])
def test_tables_bit_exact_vs_independent_restatement(nx, ny):
    got = O.Operators(nx, ny, L, L).tables()
    want = R.tables(nx, ny, L, L)
    for g, w in zip(got, want):
        assert np.array_equal(g.view(np.uint32), w.view(np.uint32))


def test_table_known_answers():
    """fftwfop.cpp:15-24,42-43,57-61 and SURVEY a1/a6."""
    N = 256
    gx, gy, lap, lapi, mask = O.Operators(N, N, L, L).tables()
    assert np.float32(np.arccos(np.float32(-1)) * np.float32(2)) == np.float32(6.2831855)
    assert gx[0] == 0 and gx[N // 2] > 0                 # Nyquist keeps a positive wavenumber
    assert np.array_equal(gx[N // 2 + 1:], -gx[1:N // 2][::-1])
    assert abs(gx[1] - 2 * np.pi / L) < 1e-11 and abs(gy[3] - 6 * np.pi / L) < 1e-11
    assert lapi[0, 0] == 1.0 and lap[0, 0] == 0.0
    assert np.array_equal(lap.ravel()[1:], lapi.ravel()[1:])
    # mask is a circle of radius sqrt(2)*ceil(N/3) = 121.6 in index space (not the 2/3 square)
    kmax2 = 2 * 86 ** 2
    assert mask[86, 86] == 0 and mask[85, 86] == 1 and mask[121, 0] == 1 and mask[122, 0] == 0
    assert mask[N - 121, 0] == 1 and mask[N - 122, 0] == 0 and mask[0, 121] == 1 and mask[0, 122] == 0
    ii = np.minimum(np.arange(N), N - np.arange(N))[:, None]
    jj = np.arange(N // 2 + 1)[None, :]
    assert np.array_equal(mask, (ii ** 2 + jj ** 2 < kmax2).astype(np.float32))


# ---------------------------------------------------------------- operators
def _rand_spec(nx, ny, seed=0):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((nx, ny // 2 + 1)) + 1j * rng.standard_normal((nx, ny // 2 + 1))).astype(np.complex64)


def test_operators_vs_fp64_fixtures():
    ops = O.Operators(64, 64, L, L)
    s = GOLD["fp64_spec_in"]
    for name, fn in (("gradx", ops.gradx), ("grady", ops.grady), ("laplacian", ops.laplacian),
                     ("invlap", ops.invertLaplacian), ("dealiase", ops.dealiase)):
        assert R.rel_l2(fn(s).view(np.float32), GOLD["fp64_" + name].view(np.float64)) < 1e-6, name


def test_operators_in_place_and_exact_forms():
    """in==out is legal (main-shallow-water.cpp:327, invert_pres.cpp:148-150)."""
    nx, ny = 96, 64
    ops = O.Operators(nx, ny, L, L)
    gx, gy, lap, lapi, mask = ops.tables()
    s = _rand_spec(nx, ny, 5)
    # exact float32 forms of fftwfop.cpp:87-124
    want = np.empty_like(s)
    want.real = -s.imag * gx[:, None]
    want.imag = s.real * gx[:, None]
    assert np.array_equal(ops.gradx(s), want)
    want.real = -s.imag * gy[None, :]
    want.imag = s.real * gy[None, :]
    assert np.array_equal(ops.grady(s), want)
    assert np.array_equal(ops.laplacian(s).view(np.float32).reshape(nx, -1, 2), s.view(np.float32).reshape(nx, -1, 2) * lap[..., None])
    assert np.array_equal(ops.invertLaplacian(s).view(np.float32).reshape(nx, -1, 2), s.view(np.float32).reshape(nx, -1, 2) / lapi[..., None])
    assert np.array_equal(ops.dealiase(s).view(np.float32).reshape(nx, -1, 2), s.view(np.float32).reshape(nx, -1, 2) * mask[..., None])
    # (0,0) mode is divided by 1, not zeroed (fftwfop.cpp:43,114)
    assert ops.invertLaplacian(s)[0, 0] == s[0, 0]
    # in-place through the C entry point
    buf = s.copy()
    p = buf.view(np.float32).ctypes.data_as(ctypes.POINTER(ctypes.c_float))
    O.lib().fbo_dealiase(ops._h, p, p)
    assert np.array_equal(buf, ops.dealiase(s))


def test_single_fourier_mode_derivatives():
    """gradx(sin(2 pi m x/L)) = (2 pi m/L) cos(...)   (SURVEY section 4 known answers)."""
    N, m, n = 64, 3, 5
    x = (np.arange(N) * (L / N))[:, None]
    y = (np.arange(N) * (L / N))[None, :]
    f = (np.sin(2 * np.pi * m * x / L) * np.cos(2 * np.pi * n * y / L)).astype(np.float32)
    ops = O.Operators(N, N, L, L)
    spec = O.r2c(f)
    dfdx = O.c2r(ops.gradx(spec), N) / (N * N)
    dfdy = O.c2r(ops.grady(spec), N) / (N * N)
    lapf = O.c2r(ops.laplacian(spec), N) / (N * N)
    kx, ky = 2 * np.pi * m / L, 2 * np.pi * n / L
    assert R.rel_l2(dfdx, kx * np.cos(kx * x) * np.cos(ky * y)) < 2e-6
    assert R.rel_l2(dfdy, -ky * np.sin(kx * x) * np.sin(ky * y)) < 2e-6
    # round-off noise in the high modes is amplified by k^2 (up to 60x the signal's k^2)
    assert R.rel_l2(lapf, -(kx ** 2 + ky ** 2) * f.astype(np.float64)) < 1e-5
    back = ops.invertLaplacian(ops.laplacian(spec))
    assert R.rel_l2(back.view(np.float32), spec.view(np.float32)) < 2e-7


# ---------------------------------------------------------------- FFT
@pytest.mark.parametrize("n", [8, 15, 64, 768, 1024, 4096, 8192, 16384])        # 8192 / 16384: the lengths the oracle checks configs 4 and 5 at
def test_fft1d_vs_numpy(n):
    rng = np.random.default_rng(n)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    X = np.fft.fft(x.astype(np.complex128))
    assert R.rel_l2(O.fft1d(x, -1).view(np.float32), X.view(np.float64)) < 3e-7
    Xi = np.fft.ifft(x.astype(np.complex128)) * n
    assert R.rel_l2(O.fft1d(x, +1).view(np.float32), Xi.view(np.float64)) < 3e-7


def test_fft1d_vs_dft_definition():
    n = 32
    rng = np.random.default_rng(7)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    k = np.arange(n)
    W = np.exp(-2j * np.pi * np.outer(k, k) / n)          # FFTW forward convention
    assert R.rel_l2(O.fft1d(x, -1).view(np.float32), (W @ x.astype(np.complex128)).view(np.float64)) < 3e-7


@pytest.mark.parametrize("nx,ny", [(64, 64), (96, 64), (256, 256), (768, 768)])
def test_r2c_c2r_vs_numpy(nx, ny):
    rng = np.random.default_rng(nx + ny)
    f = rng.standard_normal((nx, ny)).astype(np.float32)
    spec = O.r2c(f)
    assert R.rel_l2(spec.view(np.float32), np.fft.rfft2(f.astype(np.float64)).view(np.float64)) < 4e-7
    # FFT round trip  c2r(r2c(x))/GRIDS = x  (2.0e-7 at 256^2 with FFTW)
    assert R.rel_l2(O.c2r(spec, ny) / np.float32(nx * ny), f) < 5e-7
    # non-Hermitian input: SURVEY note N2 (Im at ky=0 and ky=ny/2 ignored; c2c along x first)
    s = _rand_spec(nx, ny, 3)
    want = np.fft.irfft2(s.astype(np.complex128), s=(nx, ny)) * (nx * ny)
    assert R.rel_l2(O.c2r(s, ny), want) < 4e-7


@pytest.mark.parametrize("nx,ny", [(64, 64), (96, 128), (192, 64)])
def test_r2c_c2r_vs_the_dft_definition(nx, ny):
    """The 2-D transforms against FFTW's documented DEFINITION of r2c / c2r (dense DFT matrices in float64, tests/ref_numpy.py), not
    against another FFT library: conventions (sign, no normalisation, half-spectrum layout) and the non-Hermitian c2r semantics of
    SURVEY note N2 are pinned to the mathematics.  (FFTW itself is absent from the image: DESIGN.md section 2.)"""
    rng = np.random.default_rng(7 * nx + ny)
    f = rng.standard_normal((nx, ny)).astype(np.float32)
    assert R.rel_l2(O.r2c(f).view(np.float32), R.dft2_r2c_definition(f).view(np.float64)) < 4e-7
    s = _rand_spec(nx, ny, 5)                                    # imaginary parts at ky = 0 and ky = ny/2, no Hermitian symmetry in x
    assert R.rel_l2(O.c2r(s, ny), R.dft2_c2r_definition(s, ny)) < 4e-7
    # and numpy's irfft2 has the same semantics (the yardstick of the larger tests)
    assert R.rel_l2(np.fft.irfft2(s.astype(np.complex128), s=(nx, ny)) * (nx * ny), R.dft2_c2r_definition(s, ny)) < 1e-12


def test_c2r_does_not_modify_input():
    s = _rand_spec(64, 64, 9)
    keep = s.copy()
    O.c2r(s, 64)
    assert np.array_equal(s, keep)


# ---------------------------------------------------------------- RK4 model
def test_model_fp64_fixtures_64():
    N = 64
    m = O.Model(N, N)
    v0 = GOLD["fp64_vort0"]
    assert np.array_equal(v0, O.make_field("elliptic", N))
    m.set_vort(v0)
    psi, u, v = m.diag()
    assert R.rel_l2(psi, GOLD["fp64_psi0"]) < 1e-6
    assert R.rel_l2(u, GOLD["fp64_u0"]) < 1e-6
    assert R.rel_l2(v, GOLD["fp64_v0"]) < 1e-6
    done = 0
    for upto in (1, 10, 100):
        m.step(upto - done)
        done = upto
        assert R.rel_l2(m.vort(), GOLD["fp64_vort_step%d" % upto]) < 1e-5, upto


def test_model_256_vs_fp64_100_steps():
    """config 1 grid (256^2 elliptic vortex); 100 of its 1000 steps to stay fast."""
    N = 256
    v0 = O.make_field("elliptic", N)
    m = O.Model(N, N)
    m.set_vort(v0)
    m64 = R.Model64(N, N)
    m64.set_vort(v0)
    m.step(100)
    m64.step(100)
    assert R.rel_l2(m.vort(), m64.vort()) < 1e-5


def test_model_invariants():
    """mean vorticity (the (0,0) mode) is conserved exactly; enstrophy decays."""
    N = 64
    m = O.Model(N, N)
    m.set_vort(O.make_field("elliptic", N))
    s0 = m.spectrum()
    e0 = float((m.vort().astype(np.float64) ** 2).sum())
    m.step(20)
    s1 = m.spectrum()
    assert s1[0, 0] == s0[0, 0]
    e1 = float((m.vort().astype(np.float64) ** 2).sum())
    assert e1 < e0
    # modes outside the mask circle are frozen at their initial value (SURVEY note N1)
    mask = O.Operators(N, N, L, L).tables()[4]
    assert np.array_equal(s1[mask == 0], s0[mask == 0])


def test_model_source_term():
    N = 64
    src = np.zeros((N, N), dtype=np.float32)
    O.add_cake(src, L, L, L / 2 + 50000.0, L / 2, 3e-3 / 10800.0, 30000.0)   # vort_src_input.cpp:46
    m = O.Model(N, N)
    m.set_vort(O.make_field("kuo2004", N))
    m.set_source(src)
    m64 = R.Model64(N, N)
    m64.set_vort(O.make_field("kuo2004", N))
    m64.src = src.astype(np.float64)
    m.step(10)
    m64.step(10)
    assert R.rel_l2(m.vort(), m64.vort()) < 1e-5
    m.set_source(None)


def test_rk4_fourth_order_convergence():
    """Halving dt cuts the time-stepping error ~16x (TODO.md:13 'test suite by convergence')."""
    N = 32
    v0 = O.make_field("gaussian", N)
    T = 240.0

    def run(dt):
        m = R.Model64(N, N, dt=dt)
        m.set_vort(v0)
        m.step(int(round(T / dt)))
        return m.vort()
    ref = run(0.5)
    e1 = np.linalg.norm(run(60.0) - ref)
    e2 = np.linalg.norm(run(30.0) - ref)
    assert 10.0 < e1 / e2 < 24.0


def test_oracle_under_asan_ubsan():
    """make -C oracle asan: the oracle and its self-test (oracle/selftest.c) built with -fsanitize=address,undefined
    (sanitizers run on the CPU side only; GPU ASan is unavailable on this pool)."""
    import subprocess
    d = os.path.join(os.path.dirname(HERE), "oracle")
    subprocess.check_call(["make", "-s", "-C", d, "asan"])
    res = subprocess.run([os.path.join(d, "_asan", "selftest")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         env=dict(os.environ, OMP_NUM_THREADS="2"))
    assert res.returncode == 0, res.stderr[-2000:]
    assert "selftest ok" in res.stdout and "ERROR: AddressSanitizer" not in res.stderr and "runtime error" not in res.stderr


def test_oracle_1000_steps_against_independent_fp64_run_1024():
    """The oracle's own 1000-step cross-check (BASELINE configs[1]: 1024^2 elliptic vortex, dt = 3 s): its committed fp32 result
    (oracle_1024_step1000.npz) against the independent fp64 numpy restatement on rfft2/irfft2 (fp64_1024_step1000.npz, made by
    tests/golden/make_long_fixtures.py).  SURVEY.md section 6 measured 8.3e-7 between the fp32 reference and an fp64 restatement at 256^2."""
    a = np.load(os.path.join(HERE, "golden", "oracle_1024_step1000.npz"))
    b = np.load(os.path.join(HERE, "golden", "fp64_1024_step1000.npz"))
    assert a["vort_sub4"].shape == b["vort_sub4"].shape == (256, 256)
    assert R.rel_l2(a["vort_sub4"], b["vort_sub4"]) < 5e-6
    assert abs(float(a["l2"]) / float(b["l2"]) - 1) < 5e-6
