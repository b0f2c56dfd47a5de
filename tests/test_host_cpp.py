"""C++ host side of the boundary on the GPU: the drop-in RK4 driver (host/barotropic_main.cpp,
mirror of main.cpp / main-shallow-water.cpp) and reference-shaped operator code through the
header-only shim (host/fftwfop_hip.hpp, mirror of fftwfop.hpp)."""
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
HOST = os.path.join(ROOT, "xlab-fftbarotropic_amd", "host")


def _build():
    subprocess.check_call(["make", "-s", "-C", HOST])


def test_host_programs_link():
    """CPU-side check: both host programs compile and link against the C ABI."""
    import xlab_fftbarotropic_amd as X
    X.build_lib()
    _build()
    for exe in ("barotropic_main.out", "shim_check.out", "invert_pres.out", "vort_src_input.out", "find_min.out", "makefield.out",
                "comm_bootstrap_check.out"):
        assert os.access(os.path.join(HOST, exe), os.X_OK)


@pytest.mark.gpu
def test_shim_reference_shaped_tendency(tmp_path):
    import oracle_py as O
    import ref_numpy as R
    _build()
    n = 256
    v0 = O.make_field("elliptic", n)
    (tmp_path / "input").mkdir()
    v0.tofile(str(tmp_path / "input" / "initial_vorticity.bin"))
    out = str(tmp_path / "rk1.bin")
    subprocess.check_call([os.path.join(HOST, "shim_check.out"), str(tmp_path / "input" / "initial_vorticity.bin"), out],
                          stderr=subprocess.DEVNULL)
    got = np.fromfile(out, dtype="<f4")
    ops = O.Operators(n, n, 6e5, 6e5)
    vc = O.r2c(v0)
    g = np.float32(n * n)
    dzdx = O.c2r(ops.gradx(vc), n) / g
    dzdy = O.c2r(ops.grady(vc), n) / g
    psi = ops.invertLaplacian(vc)
    u = -(O.c2r(ops.grady(psi), n) / g)
    v = O.c2r(ops.gradx(psi), n) / g
    t = -u * dzdx - v * dzdy + np.float32(0)
    tc = O.r2c(t)
    tc = (tc.view(np.float32) + ops.laplacian(vc).view(np.float32) * np.float32(6.5)).view(np.complex64)
    want = ops.dealiase(tc).view(np.float32).ravel()
    assert R.rel_l2(got, want) < 1e-5


@pytest.mark.gpu
def test_driver_outputs_match_oracle(tmp_path):
    """256^2 elliptic vortex (config 1 grid), 250 steps: files, ./log and stdout as main.cpp writes them."""
    import oracle_py as O
    import ref_numpy as R
    _build()
    n, steps = 256, 250
    (tmp_path / "in").mkdir()
    (tmp_path / "out").mkdir()
    v0 = O.make_field("elliptic", n)
    v0.tofile(str(tmp_path / "in" / "init.bin"))
    res = subprocess.run([os.path.join(HOST, "barotropic_main.out"), "-I", "in", "-O", "out", "-i", "init.bin",
                          "--npts", str(n), "--steps", str(steps)], cwd=str(tmp_path), stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, check=True, text=True)
    lines = res.stdout.splitlines()
    assert "Start project." in lines and lines[-1] == "Program ends. Congrats!"
    assert "# Step 0, time = 0.00, record now!" in lines and "# Step 1, time = 3.00" in lines
    assert "# Step 200, time = 600.00, record now!" in lines
    log = (tmp_path / "log").read_text().split()
    exp = []
    for s in (0, 100, 200):
        exp += ["out/%s_step_%d.bin" % (name, s) for name in ("vort_src_input", "vort", "psi", "u", "v")]
    assert log == exp                                            # main.cpp:160-278 order
    assert "Output out/vort_step_0.bin" in res.stderr            # fieldio.cpp:18
    rd = lambda name: np.fromfile(str(tmp_path / "out" / name), dtype="<f4").reshape(n, n)
    assert not rd("vort_src_input_step_0.bin").any()
    m = O.Model(n, n)
    m.set_vort(v0)
    for s in (0, 100, 200):
        psi, u, v = m.diag()
        assert R.rel_l2(rd("vort_step_%d.bin" % s), m.vort()) < 1e-5
        assert R.rel_l2(rd("psi_step_%d.bin" % s), psi) < 1e-5
        assert R.rel_l2(rd("u_step_%d.bin" % s), u) < 1e-5
        assert R.rel_l2(rd("v_step_%d.bin" % s), v) < 1e-5
        m.step(100)


@pytest.mark.gpu
def test_driver_optional_debug_dumps(tmp_path):
    """--dump-grad-vort / --dump-dvortdt: the OUTPUT_GRAD_VORT and OUTPUT_DVORTDT blocks of getDvortdt(debug) (main.cpp:156-162,
    170-176,229-235; not defined in the reference's configuration.hpp:4-5, which switches on OUTPUT_PSI and OUTPUT_WIND only).  With both
    on, a record step logs vort_src_input, vort, dvortdx, dvortdy, psi, u, v, dvortdt in that order; the values against the oracle's
    intermediates at 256^2 (steps 0 and 100); without the options the log is the five-file one."""
    import oracle_py as O
    import ref_numpy as R
    _build()
    n = 256
    (tmp_path / "input").mkdir()
    (tmp_path / "output").mkdir()
    v0 = O.make_field("elliptic", n)
    v0.tofile(str(tmp_path / "input" / "initial_vorticity.bin"))
    subprocess.run([os.path.join(HOST, "barotropic_main.out"), "--npts", str(n), "--steps", "101", "--dump-grad-vort", "--dump-dvortdt"],
                   cwd=str(tmp_path), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
    log = (tmp_path / "log").read_text().split()
    order = ("vort_src_input", "vort", "dvortdx", "dvortdy", "psi", "u", "v", "dvortdt")
    assert log == ["output/%s_step_%d.bin" % (name, s) for s in (0, 100) for name in order]
    rd = lambda name: np.fromfile(str(tmp_path / "output" / name), dtype="<f4").reshape(n, n)
    ops = O.Operators(n, n, 6e5, 6e5)
    g = np.float32(n * n)
    m = O.Model(n, n)
    m.set_vort(v0)
    for s_ in (0, 100):
        vc = m.spectrum()
        dzdx = O.c2r(ops.gradx(vc), n) / g                                        # main.cpp:151-154
        dzdy = O.c2r(ops.grady(vc), n) / g                                        # main.cpp:165-168
        psi, u, v = m.diag()
        t = -u * dzdx - v * dzdy + np.float32(0)                                  # main.cpp:225-227
        assert R.rel_l2(rd("dvortdx_step_%d.bin" % s_), dzdx) < 1e-5 and R.rel_l2(rd("dvortdy_step_%d.bin" % s_), dzdy) < 1e-5
        assert R.rel_l2(rd("dvortdt_step_%d.bin" % s_), t) < 1e-5
        assert R.rel_l2(rd("u_step_%d.bin" % s_), u) < 1e-5 and R.rel_l2(rd("vort_step_%d.bin" % s_), m.vort()) < 1e-5
        # the dump is the Jacobian of the dumped fields, bit for bit (the standalone kernel keeps the reference's evaluation order)
        tj = -rd("u_step_%d.bin" % s_) * rd("dvortdx_step_%d.bin" % s_) - rd("v_step_%d.bin" % s_) * rd("dvortdy_step_%d.bin" % s_) + np.float32(0)
        assert np.array_equal(rd("dvortdt_step_%d.bin" % s_), tj)
        m.step(100)
    plain = tmp_path / "plain"
    (plain / "output").mkdir(parents=True)
    subprocess.run([os.path.join(HOST, "barotropic_main.out"), "--npts", str(n), "--steps", "1", "-I", "../input"], cwd=str(plain),
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
    assert (plain / "log").read_text().split() == ["output/%s_step_0.bin" % name for name in ("vort_src_input", "vort", "psi", "u", "v")]
    multi = subprocess.run([os.path.join(HOST, "barotropic_main.out"), "--npts", str(n), "--steps", "1", "--world", "2", "--ranks-as-threads",
                            "--dump-dvortdt"], cwd=str(tmp_path), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
    assert multi.returncode == 2 and "one GPU only" in multi.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("sets", ["1", "2"])
def test_driver_fifo_source_every_step_with_records_in_flight(tmp_path, sets):
    """The FIFO source's three pinned buffers (SourceFeed) against a writer that holds sources for the records it has queued: a NEW source
    with every step (flag 1 + GRIDS float32 per step, vorticity_source.cpp:112-133), a record every 2 steps, 128^2 so that the writer is
    always behind -- with two buffer sets two queued records can hold two different old sources while a third is current and the reader
    thread wants a fourth.  The run must finish (no thread waits on another in a cycle: the writer never waits for the step loop or the
    reader), every vort_src_input_step_k.bin must be the source in force BEFORE step k's read (main-shallow-water.cpp:304 comes after the
    dump), and the last vort record must equal the Python binding fed the same sources step by step, bit for bit."""
    import threading
    import xlab_fftbarotropic_amd as X
    _build()
    n, steps, every = 128, 41, 2
    (tmp_path / "input").mkdir()
    (tmp_path / "output").mkdir()
    v0 = X.make_field("kuo2004", n)
    v0.tofile(str(tmp_path / "input" / "initial_vorticity.bin"))
    rng = np.random.default_rng(77)
    srcs = [(rng.standard_normal((n, n)) * 1e-9).astype(np.float32) for _ in range(steps)]        # srcs[k] arrives with step k's read
    fifo = str(tmp_path / "vort_src_fifo")
    os.mkfifo(fifo)
    p = subprocess.Popen([os.path.join(HOST, "barotropic_main.out"), "-f", fifo, "--npts", str(n), "--steps", str(steps), "--record-step", str(every),
                          "--record-buffers", sets], cwd=str(tmp_path), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)

    def produce():
        with open(fifo, "wb") as f:
            for k in range(steps):
                f.write(b"\x01" + srcs[k].tobytes())
    t = threading.Thread(target=produce, daemon=True)
    t.start()
    try:
        assert p.wait(timeout=180) == 0                      # a deadlock between reader, step loop and writer would end here
    finally:
        if p.poll() is None:
            p.kill()
    t.join(timeout=30)
    rd = lambda name: np.fromfile(str(tmp_path / "output" / name), dtype="<f4").reshape(n, n)
    for k in range(0, steps, every):
        want = np.zeros((n, n), dtype=np.float32) if k == 0 else srcs[k - 1]                      # dumped before step k's own read
        assert np.array_equal(rd("vort_src_input_step_%d.bin" % k), want), k
    m = X.Model(n, n)
    m.set_vort(v0)
    for k in range(steps - 1):                               # the last record is taken at step 40, before that step is computed
        m.set_source(srcs[k])
        m.step(1)
    assert np.array_equal(rd("vort_step_%d.bin" % (steps - 1)), m.vort().cpu().numpy())


@pytest.mark.gpu
def test_driver_record_path_under_pressure_one_and_two_buffer_sets(tmp_path):
    """The record path with the writer thread permanently behind: 256^2 (a step takes ~0.1 ms), a record every 3 steps, 61 steps = 21
    records of five files -- every set of pinned record buffers is always queued or being written, so the step loop sits in
    RecordWriter::acquire() and the job queue is never empty.  --record-buffers 1 (round 3's behaviour: wait for the previous record's files)
    and 2 (round 4) must give the same ./log, in main.cpp:266-282's order, and the same bytes in every file; the last record checked
    against the Python binding on the same library bit for bit (a record mixed up with a later step's copies would differ)."""
    import xlab_fftbarotropic_amd as X
    _build()
    n, steps, every = 256, 61, 3
    v0 = X.make_field("elliptic", n)
    runs = {}
    for sets in ("1", "2"):
        d = tmp_path / ("sets" + sets)
        (d / "input").mkdir(parents=True)
        (d / "output").mkdir()
        v0.tofile(str(d / "input" / "initial_vorticity.bin"))
        res = subprocess.run([os.path.join(HOST, "barotropic_main.out"), "--npts", str(n), "--steps", str(steps), "--record-step", str(every),
                              "--record-buffers", sets], cwd=str(d), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, timeout=300)
        assert res.returncode == 0, res.stderr[-1500:]
        assert ("(%s set%s)" % (sets, "" if sets == "1" else "s")) in res.stderr
        log = (d / "log").read_text().split()
        runs[sets] = (log, {f: (d / f).read_bytes() for f in log})
    want_log = ["output/%s_step_%d.bin" % (name, s) for s in range(0, steps, every) for name in ("vort_src_input", "vort", "psi", "u", "v")]
    assert runs["1"][0] == want_log and runs["2"][0] == want_log
    for f in want_log:
        assert runs["1"][1][f] == runs["2"][1][f], f
        assert len(runs["2"][1][f]) == n * n * 4
    m = X.Model(n, n)
    m.set_vort(v0)
    m.step(60)
    psi, u, v = m.diag()
    last = lambda name: np.frombuffer(runs["2"][1]["output/%s_step_60.bin" % name], dtype="<f4").reshape(n, n)
    assert np.array_equal(last("vort"), m.vort().cpu().numpy()) and np.array_equal(last("psi"), psi.cpu().numpy())
    assert np.array_equal(last("u"), u.cpu().numpy()) and np.array_equal(last("v"), v.cpu().numpy())
    mid = X.Model(n, n)                                      # ... and one from the middle of the queue
    mid.set_vort(v0)
    mid.step(30)
    assert np.array_equal(np.frombuffer(runs["2"][1]["output/vort_step_30.bin"], dtype="<f4").reshape(n, n), mid.vort().cpu().numpy())


@pytest.mark.gpu
def test_driver_4096_default_path(tmp_path):
    """The drop-in driver on the benchmark grid (4096^2 Kuo2004, dt = 0.75 s): there the engine takes its
    single-pass x transform and the digit-permutation row kernel, and the record path (get_vort / get_diag)
    converts from that path's private state layout.  Against the Python binding on the same library (bit for
    bit) and the oracle after 2 steps."""
    import oracle_py as O
    import ref_numpy as R
    import xlab_fftbarotropic_amd as X
    _build()
    n = 4096
    (tmp_path / "input").mkdir()
    (tmp_path / "output").mkdir()
    v0 = O.make_field("kuo2004", n)
    v0.tofile(str(tmp_path / "input" / "initial_vorticity.bin"))
    subprocess.run([os.path.join(HOST, "barotropic_main.out"), "--npts", str(n), "--steps", "3", "--record-step", "2",
                    "--dt", "0.75"], cwd=str(tmp_path), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
    rd = lambda name: np.fromfile(str(tmp_path / "output" / name), dtype="<f4").reshape(n, n)
    m = X.Model(n, n, dt=0.75)
    m.set_vort(v0)
    m.step(2)
    psi, u, v = m.diag()
    assert np.array_equal(rd("vort_step_2.bin"), m.vort().cpu().numpy())
    assert np.array_equal(rd("psi_step_2.bin"), psi.cpu().numpy())
    assert np.array_equal(rd("u_step_2.bin"), u.cpu().numpy())
    assert np.array_equal(rd("v_step_2.bin"), v.cpu().numpy())
    mo = O.Model(n, n, dt=0.75)
    mo.set_vort(v0)
    mo.step(2)
    po, uo, vo = mo.diag()
    assert R.rel_l2(rd("vort_step_2.bin"), mo.vort()) < 1e-5
    assert R.rel_l2(rd("u_step_2.bin"), uo) < 1e-5 and R.rel_l2(rd("psi_step_2.bin"), po) < 1e-5


@pytest.mark.gpu
def test_driver_fifo_source(tmp_path):
    """main-shallow-water.cpp path: per step one flag byte, GRIDS float32 after a flag of 1
    (vorticity_source.cpp:112-133); the producer closes early so the last reads hit EOF."""
    import oracle_py as O
    import ref_numpy as R
    _build()
    n, steps = 128, 12
    (tmp_path / "input").mkdir()
    (tmp_path / "output").mkdir()
    v0 = O.make_field("kuo2004", n)
    v0.tofile(str(tmp_path / "input" / "initial_vorticity.bin"))
    src = np.zeros((n, n), dtype=np.float32)
    O.add_cake(src, 6e5, 6e5, 6e5 / 2 + 5e4, 6e5 / 2, 3e-3 / 10800.0, 3e4)
    stream = b"\x00" * 3 + b"\x01" + src.tobytes() + b"\x00" * 4 + b"\x01" + np.zeros_like(src).tobytes() + b"\x00"
    fifo = str(tmp_path / "vort_src_fifo")
    os.mkfifo(fifo)
    p = subprocess.Popen([os.path.join(HOST, "barotropic_main.out"), "-f", fifo, "--npts", str(n), "--steps", str(steps),
                          "--record-step", "6"], cwd=str(tmp_path), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    with open(fifo, "wb") as f:
        f.write(stream)
    assert p.wait(timeout=120) == 0
    m = O.Model(n, n)
    m.set_vort(v0)
    for step in range(steps):
        if step == 6:
            got = np.fromfile(str(tmp_path / "output" / "vort_step_6.bin"), dtype="<f4").reshape(n, n)
            assert R.rel_l2(got, m.vort()) < 1e-5
            s6 = np.fromfile(str(tmp_path / "output" / "vort_src_input_step_6.bin"), dtype="<f4").reshape(n, n)
            assert np.array_equal(s6, src)                       # dumped before this step's read, source switched on at step 3
        if step == 3:
            m.set_source(src)
        if step == 8:
            m.set_source(None)
        m.step(1)


@pytest.mark.gpu
def test_invert_pres_against_oracle_pipeline(tmp_path):
    """invert_pres.cpp:114-188 on the C ABI, driven exactly like test/02-test_invert_pressure: the
    driver's ./log, rewritten psi -> pres, piped into invert_pres."""
    import re
    import oracle_py as O
    import ref_numpy as R
    _build()
    n = 256
    (tmp_path / "input").mkdir()
    (tmp_path / "output").mkdir()
    O.make_field("kuo2004", n).tofile(str(tmp_path / "input" / "initial_vorticity.bin"))
    subprocess.check_call([os.path.join(HOST, "barotropic_main.out"), "--npts", str(n), "--steps", "101"], cwd=str(tmp_path),
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    lines = []
    for ln in (tmp_path / "log").read_text().split():                    # the perl one-liner of example.sh:15
        mm = re.match(r"(.*/)psi(.*?\.bin)", ln)
        if mm:
            lines.append("%s=>%spres%s" % (ln, mm.group(1), mm.group(2)))
    assert len(lines) == 2
    res = subprocess.run([os.path.join(HOST, "invert_pres.out"), "--npts", str(n), "-x", "3", "-y", "5"], cwd=str(tmp_path),
                         input="\n".join(lines) + "\n", stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, check=True)
    assert res.stdout.strip().endswith("Program ends. Congrats!")
    ops = O.Operators(n, n, 6e5, 6e5)
    g = np.float32(n * n)
    for step in (0, 100):
        psi = np.fromfile(str(tmp_path / "output" / ("psi_step_%d.bin" % step)), dtype="<f4").reshape(n, n)
        pc = O.r2c(psi)
        c2 = lambda s: O.c2r(ops.dealiase(s), n) / g
        ty = ops.grady(pc)
        dx2, dy2, dxdy = c2(ops.gradx(ops.gradx(pc))), c2(ops.grady(ty)), c2(ops.gradx(ty))
        curv = dx2 * dy2 - dxdy * dxdy
        lp = O.r2c(curv).view(np.float32)
        lp = (lp + lp) + ops.laplacian(pc).view(np.float32) * np.float32(1e-5)
        pres = O.c2r(ops.invertLaplacian(lp.view(np.complex64)), n) / g
        pres = pres - pres.ravel()[3 + n * 5]
        got = np.fromfile(str(tmp_path / "output" / ("pres_step_%d.bin" % step)), dtype="<f4").reshape(n, n)
        # the same pipeline in fp64 numpy (tables of the reference's float32 values): how far is ANY float32 evaluation from it?
        gx, gy, lap, lapi, mask = (t.astype(np.float64) for t in R.tables(n, n, 6e5, 6e5))
        p64 = np.fft.rfft2(psi.astype(np.float64))
        ikx, iky = 1j * gx[:, None], 1j * gy[None, :]
        i2 = lambda s_: np.fft.irfft2(s_ * mask, s=(n, n))
        curv64 = i2(ikx * ikx * p64) * i2(iky * iky * p64) - i2(ikx * iky * p64) ** 2
        pres64 = np.fft.irfft2((lap * p64 * 1e-5 + 2.0 * np.fft.rfft2(curv64)) / lapi, s=(n, n))
        pres64 = pres64 - pres64.ravel()[3 + n * 5]
        e_gpu_oracle, e_oracle_64, e_gpu_64 = R.rel_l2(got, pres), R.rel_l2(pres, pres64), R.rel_l2(got, pres64)
        print("invert_pres step %d: GPU vs oracle %.2e, oracle vs fp64 %.2e, GPU vs fp64 %.2e" % (step, e_gpu_oracle, e_oracle_64, e_gpu_64))
        # measured on MI355X: GPU vs oracle 7e-7 .. 1e-6, oracle vs fp64 4e-7 .. 5e-7, GPU vs fp64 8e-7 .. 1e-6 -- the pipeline
        # (second derivatives, their product, a Laplacian inversion, one reference point subtracted from every value) does not
        # amplify float32 rounding beyond the bar of the RK4 path, so the same bar applies (round 1 asserted 1e-4 here)
        assert e_gpu_oracle < 1e-5 and e_gpu_64 < 1e-5 and e_oracle_64 < 1e-5, step


def test_fifo_producer_stream_matches_reference_and_oracle():
    """host/vort_src_input.cpp against (a) the reference-built producer's byte stream at its compiled-in
    configuration (768^2, dt = 3, 1200 steps: all flags 0 because beg_step = 2400 > 1200) and (b) the
    oracle's cake when the window is inside the run.  Host-only: runs without a GPU."""
    import hashlib
    import json
    import oracle_py as O
    _build()
    meta = json.load(open(os.path.join(HERE, "golden", "ref_meta.json")))
    exe = os.path.join(HOST, "vort_src_input.out")
    raw = subprocess.run([exe, "--npts", "768", "--dt", "3", "--steps", "1200"], stdout=subprocess.PIPE,
                         stderr=subprocess.DEVNULL, check=True).stdout
    assert hashlib.sha256(raw).hexdigest() == meta["fifo_stream"]["sha256"] and len(raw) == 1199
    n = 64
    raw = subprocess.run([exe, "--npts", str(n), "--dt", "3", "--steps", "12", "--beg-time", "9", "--duration", "12"],
                         stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
    src = np.zeros((n, n), dtype=np.float32)
    O.add_cake(src, 6e5, 6e5, 6e5 / 2 + 5e4, 6e5 / 2, np.float32(3e-3 / np.float32(12.0)), 3e4)
    want = b"\x00" * 2 + b"\x01" + src.tobytes() + b"\x00" * 3 + b"\x01" + np.zeros_like(src).tobytes() + b"\x00" * 4
    assert raw == want                      # steps 1..11: on at step 3 (= 9 s / 3 s), off at step 7


@pytest.mark.gpu
def test_driver_restart_with_start_step(tmp_path):
    """--start-step: restarting from vort_step_100.bin continues numbering and reproduces the
    uninterrupted run's later dumps to rounding (the restart goes through one extra c2r/r2c pair)."""
    import oracle_py as O
    import ref_numpy as R
    _build()
    n = 256
    for d in ("a", "b"):
        (tmp_path / d / "input").mkdir(parents=True)
        (tmp_path / d / "output").mkdir()
    O.make_field("elliptic", n).tofile(str(tmp_path / "a" / "input" / "initial_vorticity.bin"))
    exe = os.path.join(HOST, "barotropic_main.out")
    run = lambda cwd, extra: subprocess.check_call([exe, "--npts", str(n)] + extra, cwd=str(tmp_path / cwd),
                                                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    run("a", ["--steps", "201"])
    os.link(str(tmp_path / "a" / "output" / "vort_step_100.bin"), str(tmp_path / "b" / "input" / "vort_step_100.bin"))
    run("b", ["-i", "vort_step_100.bin", "--start-step", "100", "--steps", "201"])
    assert sorted(os.listdir(str(tmp_path / "b" / "output"))) == sorted(
        "%s_step_%d.bin" % (nm, s) for s in (100, 200) for nm in ("vort_src_input", "vort", "psi", "u", "v"))
    rd = lambda d, f: np.fromfile(str(tmp_path / d / "output" / f), dtype="<f4")
    assert R.rel_l2(rd("b", "vort_step_200.bin"), rd("a", "vort_step_200.bin")) < 2e-6


# ---------------------------------------------------------------------------------------------------
# boundary fidelity: the reference's own link names (fieldio.hpp:5-6, <fftw3.h>) resolved by the product
# ---------------------------------------------------------------------------------------------------
LIBDIR = os.path.join(ROOT, "xlab-fftbarotropic_amd", "lib")
REFDIR = os.path.join(ROOT, "oracle", "_ref")

_FIELDIO_CHILD = r"""
import ctypes, sys, numpy as np
lib = ctypes.CDLL(sys.argv[1])
wr, rd = getattr(lib, "_Z10writeFieldPKcPfm"), getattr(lib, "_Z9readFieldPKcPfm")
for fn in (wr, rd):
    fn.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_float), ctypes.c_size_t]; fn.restype = None
a = np.random.default_rng(7).standard_normal(12345).astype(np.float32)
fp = lambda x: x.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
wr(b"field.bin", fp(a), a.size)
b = np.zeros_like(a); rd(b"field.bin", fp(b), b.size)
assert np.array_equal(a, b)
short = np.zeros(20000, np.float32); rd(b"field.bin", fp(short), short.size)      # short read: fieldio.cpp:26 stays silent
"""


@pytest.mark.skipif(not os.path.exists(os.path.join(REFDIR, "libfieldio.so")), reason="oracle/_ref not built")
def test_product_fieldio_library_matches_reference_library(tmp_path):
    """lib/libfieldio.so (csrc/fb_fieldio.cpp) exports the reference's mangled writeField/readField (fieldio.hpp:5-6)
    and behaves byte for byte like the reference's own fieldio.cpp compiled into oracle/_ref: same file, same
    stderr lines, including the short read that reports the element count as "bytes" (fieldio.cpp:32)."""
    import sys
    import xlab_fftbarotropic_amd as X
    X.build_lib()
    outs = {}
    for tag, lib in (("ref", os.path.join(REFDIR, "libfieldio.so")), ("mine", os.path.join(LIBDIR, "libfieldio.so")),
                     ("abi", os.path.join(LIBDIR, "libfftbaro.so"))):
        d = tmp_path / tag
        d.mkdir()
        res = subprocess.run([sys.executable, "-c", _FIELDIO_CHILD, lib], cwd=str(d), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                             check=True)
        outs[tag] = (open(str(d / "field.bin"), "rb").read(), res.stderr)
    assert outs["mine"][0] == outs["ref"][0] and outs["abi"][0] == outs["ref"][0] and len(outs["ref"][0]) == 4 * 12345
    assert outs["mine"][1] == outs["ref"][1] == b"Output field.bin\n12345 bytes read: field.bin\n12345 bytes read: field.bin\n"
    assert outs["abi"][1] == outs["ref"][1]


def test_fftw_shim_exports_every_declared_name():
    """lib/libfftw3f_fb.so exports the FFTW3 names include/fftw3_fb.h declares (main.cpp:103-135,154)."""
    import ctypes
    import re
    import xlab_fftbarotropic_amd as X
    X.lib()                                              # libfftbaro.so (and torch's HIP runtime) first
    src = open(os.path.join(ROOT, "include", "fftw3_fb.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = sorted(set(re.findall(r"\b(fftwf_[a-z0-9_]+)\s*\(", src)))
    assert {"fftwf_malloc", "fftwf_free", "fftwf_plan_dft_r2c_2d", "fftwf_plan_dft_c2r_2d", "fftwf_execute", "fftwf_destroy_plan"} <= set(names)
    L = ctypes.CDLL(os.path.join(LIBDIR, "libfftw3f_fb.so"))
    assert not [n for n in names if not hasattr(L, n)]


@pytest.mark.skipif(not os.path.exists(os.path.join(REFDIR, "find_min.out")), reason="oracle/_ref not built")
def test_find_min_matches_reference_build(tmp_path):
    """host/find_min.cpp against the reference's find_min.cpp:67-101 compiled from its own source (grid fixed at
    768^2 there): same 30 lines in the same (replacement) order, ties and all; also under ASan/UBSan."""
    import oracle_py as O
    _build()
    subprocess.check_call(["make", "-s", "-C", HOST, "asan"])
    n = 768
    rng = np.random.default_rng(3)
    f = (O.make_field("kuo2004", n) * np.float32(-1e3)).astype(np.float32)
    f += rng.integers(0, 4, size=(n, n)).astype(np.float32) * np.float32(1e-3)          # many exact ties
    names = []
    for i in range(2):
        (f + np.float32(i)).astype(np.float32).tofile(str(tmp_path / ("pres_%d.bin" % i)))
        names.append("pres_%d.bin" % i)
    inp = "\n".join(names) + "\n"
    run = lambda cmd: subprocess.run(cmd, cwd=str(tmp_path), input=inp, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, check=True)
    ref = run([os.path.join(REFDIR, "find_min.out")])
    mine = run([os.path.join(HOST, "find_min.out"), "--npts", str(n)])
    asan = run([os.path.join(HOST, "asan", "find_min.out"), "--npts", str(n)])
    assert mine.stdout == ref.stdout and len(ref.stdout.splitlines()) == 60
    assert mine.stderr == ref.stderr
    assert asan.stdout == ref.stdout and "ERROR" not in asan.stderr and "runtime error" not in asan.stderr
    ix, iy, val = ref.stdout.splitlines()[0].split()
    assert float(val) == pytest.approx(float(f[int(ix), int(iy)]), rel=1e-5)


def test_fifo_producer_under_asan():
    """host/vort_src_input.cpp + csrc/fb_fields.cpp under AddressSanitizer/UBSan (GPU ASan is unavailable on this pool)."""
    subprocess.check_call(["make", "-s", "-C", HOST, "asan"])
    raw = subprocess.run([os.path.join(HOST, "asan", "vort_src_input.out"), "--npts", "64", "--dt", "3", "--steps", "12", "--beg-time", "9",
                          "--duration", "12"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
    assert len(raw.stdout) == 11 + 2 * 64 * 64 * 4 and b"ERROR" not in raw.stderr and b"runtime error" not in raw.stderr


def test_makefield_program_matches_the_reference_generators(tmp_path):
    """host/makefield.out (fb_make_field + fb_write_field; makefield-elliptic-vortex.cpp:12-58, makefield-Kuo2004.cpp:30-41,
    makefield-gaussian.cpp:14-31, makefield-const-vortex.cpp:14-38): at the reference's compiled-in NPTS = 768 the files carry the
    hashes of the reference-built generators' output (tests/golden/ref_meta.json, made from oracle/_ref), both through --kind and
    when started under the reference's program names; stderr is writeField's line (fieldio.cpp:18).  Host only (no GPU call).  Also
    under AddressSanitizer + UBSan."""
    import hashlib
    import json
    import xlab_fftbarotropic_amd as X
    X.build_lib()
    _build()
    subprocess.check_call(["make", "-s", "-C", HOST, "asan"])
    meta = json.load(open(os.path.join(HERE, "golden", "ref_meta.json")))
    (tmp_path / "input").mkdir()
    (tmp_path / "alt").mkdir()
    names = {"elliptic": "makefield-elliptic-vortex.out", "kuo2004": "makefield-Kuo2004.out", "gaussian": "makefield-gaussian.out",
             "const": "makefield-const-vortex.out"}
    for kind, prog in names.items():
        res = subprocess.run([os.path.join(HOST, "makefield.out"), "--kind", kind], cwd=str(tmp_path), stdout=subprocess.PIPE,
                             stderr=subprocess.PIPE, text=True, check=True)
        assert res.stderr == "Output input/initial_vorticity.bin\n" and res.stdout == ""
        raw = (tmp_path / "input" / "initial_vorticity.bin").read_bytes()
        assert len(raw) == meta[kind]["nbytes"] and hashlib.sha256(raw).hexdigest() == meta[kind]["sha256"], kind
        link = tmp_path / prog                                                    # the reference's program name, no arguments
        os.symlink(os.path.join(HOST, "makefield.out"), str(link))
        subprocess.run([str(link), "-I", "alt", "-i", "f.bin"], cwd=str(tmp_path), stderr=subprocess.DEVNULL, check=True)
        assert (tmp_path / "alt" / "f.bin").read_bytes() == raw
        if os.path.exists(os.path.join(REFDIR, prog)):                            # build container: the reference's own program, side by side
            (tmp_path / "ref" / "input").mkdir(parents=True, exist_ok=True)
            subprocess.run([os.path.join(REFDIR, prog)], cwd=str(tmp_path / "ref"), stderr=subprocess.DEVNULL, check=True)
            assert (tmp_path / "ref" / "input" / "initial_vorticity.bin").read_bytes() == raw
    san = subprocess.run([os.path.join(HOST, "asan", "makefield.out"), "--kind", "kuo2004", "--npts", "192", "-I", "alt", "-i", "s.bin"],
                         cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, check=True)
    assert "ERROR" not in san.stderr and "runtime error" not in san.stderr
    small = subprocess.run([os.path.join(HOST, "makefield.out"), "--kind", "kuo2004", "--npts", "192", "-I", "alt", "-i", "t.bin"],
                           cwd=str(tmp_path), stderr=subprocess.DEVNULL)
    assert small.returncode == 0 and (tmp_path / "alt" / "s.bin").read_bytes() == (tmp_path / "alt" / "t.bin").read_bytes()
    assert subprocess.run([os.path.join(HOST, "makefield.out")], cwd=str(tmp_path), stderr=subprocess.DEVNULL).returncode == 2
    assert subprocess.run([os.path.join(HOST, "makefield.out"), "--kind", "nope"], cwd=str(tmp_path), stderr=subprocess.DEVNULL).returncode == 1


@pytest.mark.gpu
def test_runtest_example_on_product_binaries_only(tmp_path):
    """test/01-runtest/example.sh:3-10 with nothing but product binaries: `mkdir input output`, the generator under the reference's
    program name (a link to host/makefield.out), then the driver with no arguments (NPTS = 768, dt = 3 s, 1200 steps, a record every 100:
    configuration.hpp:18,34-36).  Checked: the step lines and ./log as main.cpp:262-264,266-282 writes them, the records against the oracle
    at steps 0 and 100, and the driver's [timing] summary on stderr (stdout carries none of it)."""
    import re
    import oracle_py as O
    import ref_numpy as R
    _build()
    n = 768
    (tmp_path / "input").mkdir()
    (tmp_path / "output").mkdir()
    os.symlink(os.path.join(HOST, "makefield.out"), str(tmp_path / "makefield-elliptic-vortex.out"))
    subprocess.run([str(tmp_path / "makefield-elliptic-vortex.out")], cwd=str(tmp_path), stderr=subprocess.DEVNULL, check=True)
    res = subprocess.run([os.path.join(HOST, "barotropic_main.out")], cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    steps = [ln for ln in res.stdout.splitlines() if ln.startswith("# Step")]
    assert len(steps) == 1200 and steps[100] == "# Step 100, time = 300.00, record now!" and steps[1199] == "# Step 1199, time = 3597.00"
    assert "[timing]" not in res.stdout and res.stdout.splitlines()[-1] == "Program ends. Congrats!"
    log = (tmp_path / "log").read_text().split()
    assert len(log) == 12 * 5 and log[:2] == ["output/vort_src_input_step_0.bin", "output/vort_step_0.bin"]
    tl = [ln for ln in res.stderr.splitlines() if ln.startswith("[timing]")]
    assert len(tl) == 4 and tl[3].startswith("[timing] host time inside the 12 record branches:")
    m3 = re.search(r"of those ([0-9.]+) s: ([0-9.]+) s stepping, ([0-9.]+) s with a record step holding the compute stream .*?, ([0-9.]+) s between the last "
                   r"step's end and the last file \(writer tail\), (-?[0-9.]+) s unaccounted", tl[2])
    assert m3, tl[2]
    parts = [float(m3.group(i)) for i in range(1, 6)]
    assert abs(parts[0] - sum(parts[1:])) < 2e-3 and all(x >= -2e-3 for x in parts[1:4])      # the four parts add up to the wall time
    m1 = re.search(r"1200 RK4 steps, 768 x 768 grid, 1 GPU: step loop without the record steps ([0-9.]+) steps/s \(([0-9.]+) ms/step\) = ([0-9.]+) GB/s "
                   r"by 320 N\^2 B/step = ([0-9.]+) of 8000 GB/s", tl[0])
    m2 = re.search(r"with 12 record steps \(([0-9.]+) GB written\): ([0-9.]+) steps/s over ([0-9.]+) s of wall time", tl[1])
    assert m1 and m2, tl
    rate, wall_rate = float(m1.group(1)), float(m2.group(2))
    assert rate > 0 and wall_rate > 0 and wall_rate <= rate * 1.02
    assert abs(float(m1.group(3)) - 320.0 * n * n * rate / 1e9) <= 1.0 and abs(float(m2.group(1)) - 12 * 5 * n * n * 4 / 1e9) < 1e-3
    v0 = np.fromfile(str(tmp_path / "input" / "initial_vorticity.bin"), dtype="<f4").reshape(n, n)
    assert np.array_equal(v0, O.make_field("elliptic", n))
    m = O.Model(n, n)
    m.set_vort(v0)
    rd = lambda name: np.fromfile(str(tmp_path / "output" / name), dtype="<f4").reshape(n, n)
    for s_ in (0, 100):
        psi, u, v = m.diag()
        assert R.rel_l2(rd("vort_step_%d.bin" % s_), m.vort()) < 1e-5 and R.rel_l2(rd("psi_step_%d.bin" % s_), psi) < 1e-5
        assert R.rel_l2(rd("u_step_%d.bin" % s_), u) < 1e-5 and R.rel_l2(rd("v_step_%d.bin" % s_), v) < 1e-5
        m.step(100)
    quiet = subprocess.run([os.path.join(HOST, "barotropic_main.out"), "--steps", "3", "--no-timing"], cwd=str(tmp_path), stdout=subprocess.DEVNULL,
                           stderr=subprocess.PIPE, text=True, check=True)
    assert "[timing]" not in quiet.stderr


@pytest.mark.gpu
def test_record_files_have_the_plotting_layout(tmp_path):
    """test/01-runtest/plot/draw_figs.py:103-105 reads every dump as np.fromfile('<f4', count=nx*ny).reshape((nx, ny)).transpose():
    index [i = x][j = y] on disk (configuration.hpp:31), image rows = y after the transpose.  Kuo2004's weak vortex sits
    50 km to the +x side of the strong one (makefield-Kuo2004.cpp:35-38), which only shows up on the right axis."""
    _build()
    n = 256
    import oracle_py as O
    (tmp_path / "input").mkdir()
    (tmp_path / "output").mkdir()
    O.make_field("kuo2004", n).tofile(str(tmp_path / "input" / "initial_vorticity.bin"))
    subprocess.check_call([os.path.join(HOST, "barotropic_main.out"), "--npts", str(n), "--steps", "1"], cwd=str(tmp_path),
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    L, dx = 6e5, 6e5 / n
    for name in ("vort", "u", "v", "psi", "vort_src_input"):
        assert os.path.getsize(str(tmp_path / "output" / ("%s_step_0.bin" % name))) == 4 * n * n
    img = np.fromfile(str(tmp_path / "output" / "vort_step_0.bin"), dtype="<f4", count=n * n).reshape((n, n)).transpose()
    iy, ix = np.unravel_index(np.argmax(img), img.shape)                  # image[row = y][col = x]
    assert abs(ix - (L / 2) / dx) <= 1 and abs(iy - (L / 2) / dx) <= 1     # strong vortex at the centre
    i_weak = int(round((L / 2 + 5e4) / dx))
    assert img[n // 2, i_weak] > 2.5e-3 and img[i_weak, n // 2] < 1e-4      # weak vortex: +x of the centre, not +y
    u = np.fromfile(str(tmp_path / "output" / "u_step_0.bin"), dtype="<f4", count=n * n).reshape((n, n)).transpose()
    c, r = n // 2, int(round(2e4 / dx))
    # u = -dpsi/dy: both vortices sit on the line y = L/2, so u is antisymmetric ACROSS image rows (westward north of the
    # cyclones, eastward south of them) and vanishes along the row through the centres
    assert u[c + r, c] < -5 and u[c - r, c] > 5 and abs(u[c + r, c] + u[c - r, c]) < 1e-2
    assert abs(u[c, c + r]) < 1e-2 and abs(u[c, c - r]) < 1e-2


def test_fifo_producer_emits_one_ranks_rows():
    """vort_src_input.out --world P --rank r (multi-GPU runs feed one FIFO per rank): flags as usual, fields cut to that rank's rows."""
    _build()
    exe = os.path.join(HOST, "vort_src_input.out")
    n = 64
    args = ["--npts", str(n), "--dt", "3", "--steps", "12", "--beg-time", "9", "--duration", "12"]
    full = subprocess.run([exe] + args, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
    fld = np.frombuffer(full[3:3 + 4 * n * n], dtype="<f4").reshape(n, n)
    assert fld.max() > 0
    for r in range(4):
        part = subprocess.run([exe] + args + ["--world", "4", "--rank", str(r)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
        rows = fld[r * 16:(r + 1) * 16].tobytes()
        assert part == b"\x00" * 2 + b"\x01" + rows + b"\x00" * 3 + b"\x01" + bytes(len(rows)) + b"\x00" * 4


@pytest.mark.gpu
def test_driver_four_ranks_as_threads_match_the_single_gpu_run(tmp_path):
    """barotropic_main.out --world 4 --ranks-as-threads (all ranks of the multi-GPU flow inside one process on this one GPU:
    per-rank row ranges of the input and record files, per-rank FIFO sources, engine-driven transposes) against the
    single-GPU run of the same configuration: same ./log, same step lines, record files equal (vort bit for bit)."""
    import oracle_py as O
    import ref_numpy as R
    _build()
    n, steps, world = 512, 201, 4
    exe, prod = os.path.join(HOST, "barotropic_main.out"), os.path.join(HOST, "vort_src_input.out")
    src_args = ["--npts", str(n), "--dt", "3", "--steps", str(steps), "--beg-time", "150", "--duration", "300"]   # on at step 50, off at 150
    outs = {}
    for tag in ("one", "four", "fan"):
        d = tmp_path / tag
        (d / "input").mkdir(parents=True)
        (d / "output").mkdir()
        O.make_field("kuo2004", n).tofile(str(d / "input" / "initial_vorticity.bin"))
        procs = []
        if tag == "one":
            os.mkfifo(str(d / "fifo"))
            procs.append(subprocess.Popen("%s %s > fifo" % (prod, " ".join(src_args)), shell=True, cwd=str(d), stderr=subprocess.DEVNULL))
            extra = []
        elif tag == "fan":                                  # ONE producer of whole fields (what the reference ships); the lead rank fans its records out
            os.mkfifo(str(d / "fifo"))
            for r in range(world):
                os.mkfifo(str(d / ("fifo.%d" % r)))
            procs.append(subprocess.Popen("%s %s > fifo" % (prod, " ".join(src_args)), shell=True, cwd=str(d), stderr=subprocess.DEVNULL))
            extra = ["--world", str(world), "--ranks-as-threads", "--fifo-fanout"]
        else:
            for r in range(world):
                os.mkfifo(str(d / ("fifo.%d" % r)))
                procs.append(subprocess.Popen("%s %s --world %d --rank %d > fifo.%d" % (prod, " ".join(src_args), world, r, r), shell=True,
                                              cwd=str(d), stderr=subprocess.DEVNULL))
            extra = ["--world", str(world), "--ranks-as-threads"]
        res = subprocess.run([exe, "--npts", str(n), "--steps", str(steps), "-f", "fifo"] + extra, cwd=str(d), stdout=subprocess.PIPE,
                             stderr=subprocess.PIPE, text=True, timeout=600)
        for p in procs:
            p.wait(timeout=60)
        assert res.returncode == 0, res.stderr[-2000:]
        outs[tag] = (res.stdout, (d / "log").read_text())
    assert outs["four"][0] == outs["one"][0] and outs["four"][1] == outs["one"][1]
    assert outs["fan"][0] == outs["one"][0] and outs["fan"][1] == outs["one"][1]
    assert "# Step 200, time = 600.00, record now!" in outs["one"][0] and len(outs["one"][1].split()) == 15
    rd = lambda tag, f: np.fromfile(str(tmp_path / tag / "output" / f), dtype="<f4")
    for step in (0, 100, 200):
        a, b = rd("one", "vort_step_%d.bin" % step), rd("four", "vort_step_%d.bin" % step)
        assert a.size == n * n and np.array_equal(a.view(np.uint32), b.view(np.uint32)), step
        assert np.array_equal(rd("one", "vort_src_input_step_%d.bin" % step), rd("four", "vort_src_input_step_%d.bin" % step))
        for name in ("psi", "u", "v"):
            assert R.rel_l2(rd("four", "%s_step_%d.bin" % (name, step)), rd("one", "%s_step_%d.bin" % (name, step))) < 1e-6, (name, step)
        for f in ("vort", "psi", "u", "v", "vort_src_input"):                       # fanned-out source == per-rank producers, bit for bit
            assert np.array_equal(rd("fan", "%s_step_%d.bin" % (f, step)).view(np.uint32), rd("four", "%s_step_%d.bin" % (f, step)).view(np.uint32)), (f, step)
    assert rd("one", "vort_src_input_step_100.bin").max() > 0 and rd("one", "vort_src_input_step_200.bin").max() == 0


def test_comm_file_bootstrap_never_takes_a_stale_id(tmp_path):
    """host/comm_bootstrap.hpp (the --comm-file hand-over of barotropic_main.out --world P --rank r; VERDICT r2 / ADVICE r2):
    a leftover record of a previous launch must not reach ncclCommInitRank.  No GPU: comm_bootstrap_check.out is the header's
    command-line face."""
    import time
    _build()
    exe = os.path.join(HOST, "comm_bootstrap_check.out")
    f = str(tmp_path / "comm")
    run = lambda *a, **k: subprocess.run([exe] + [str(x) for x in a], stdout=subprocess.PIPE, text=True, timeout=60, **k)
    # 1. tokens: a stale record of launch A lies there; the reader of launch B waits for B's record
    assert run("publish", f, "launch-A", 17).returncode == 0
    rd = subprocess.Popen([exe, "await", f, "launch-B", "20"], stdout=subprocess.PIPE, text=True)
    time.sleep(0.5)
    assert rd.poll() is None                                            # still waiting: A's record was not taken
    assert run("publish", f, "launch-B", 42).returncode == 0
    assert rd.communicate(timeout=30)[0].strip() == "id 42" and rd.returncode == 0
    # 2. rank 0 removes whatever lies at the path before anything else
    assert run("prepare", f).returncode == 0 and not os.path.exists(f)
    r = run("await", f, "launch-B", "1")
    assert r.stdout.strip() == "timeout" and r.returncode == 3
    # 3. no token: only a fresh record counts (older than max-age at the reader's start -> skipped), a fresh one is taken
    assert run("publish", f, "", 5).returncode == 0
    old = time.time() - 1000
    os.utime(f, (old, old))
    assert run("await", f, "", "1", "60").stdout.strip() == "timeout"
    assert run("publish", f, "", 6).returncode == 0
    assert run("await", f, "", "5", "60").stdout.strip() == "id 6"
    # 4. a foreign or truncated file is not a record
    open(f, "wb").write(b"\0" * 128)                                    # round 2's format: a bare id
    assert run("await", f, "", "1").stdout.strip() == "timeout"
    # 5. an over-long token is refused
    assert run("publish", f, "x" * 64, 1).returncode == 1


LINKDIR = os.path.join(REFDIR, "linkcheck")


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="the reference tree is only present in the build container")
def test_reference_sources_link_unmodified_against_the_boundary():
    """the reference's own main.cpp, main-shallow-water.cpp and invert_pres.cpp -- unmodified, compiled where they
    lie -- build with -I host/compat -I include and link with -lfftw3f_fb -lfieldio (oracle/Makefile `linkcheck`, outputs under
    oracle/_ref/linkcheck/).  `#include <fftw3.h>` (main.cpp:12), `#include "fftwfop.cpp"` (main.cpp:19: the reference's own operator
    class, on fftwf_malloc'ed memory) and the six FFTW entry points (main.cpp:103-135,154) resolve to the engine.  A link check of the
    boundary, not an oracle: the FFT inside these binaries is the engine's."""
    import xlab_fftbarotropic_amd as X
    X.build_lib()
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "linkcheck"])
    for prog in ("main.out", "main-shallow-water.out", "invert_pres.out"):
        exe = os.path.join(LINKDIR, prog)
        assert os.access(exe, os.X_OK), prog
        dyn = subprocess.run(["readelf", "-d", exe], stdout=subprocess.PIPE, text=True, check=True).stdout
        assert "libfftw3f_fb.so" in dyn and "libfieldio.so" in dyn and "libfftw3f.so" not in dyn.replace("libfftw3f_fb", "")
        syms = subprocess.run(["nm", "-D", "--undefined-only", exe], stdout=subprocess.PIPE, text=True, check=True).stdout
        for name in ("fftwf_malloc", "fftwf_plan_dft_r2c_2d", "fftwf_plan_dft_c2r_2d", "fftwf_execute"):
            assert name in syms, (prog, name)
        assert "_Z10writeFieldPKcPfm" in syms or "_Z9readFieldPKcPfm" in syms, prog       # fieldio.hpp:5-6 by their C++ names


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(os.path.join(REFDIR, "linkcheck", "main-shallow-water.out")), reason="oracle/_ref/linkcheck not built")
def test_reference_pipeline_unmodified_runs_on_the_engine(tmp_path):
    """The reference's own test (test/02-test_invert_pressure/example.sh) with the reference's own programs, none of them modified:
    makefield-Kuo2004.out, vort_src_input.out and find_min.out as built into oracle/_ref from their sources, and main-shallow-water.out
    and invert_pres.out built from theirs against the engine's FFTW-named boundary (oracle/_ref/linkcheck).  NPTS = 768, dt = 3 s,
    1200 steps, the source switched on and off by the producer -- the configuration the reference ships.  Beside it the engine's own
    driver (host/barotropic_main.out, fused path, same FIFO protocol) runs the same pipeline.  Compared: stdout's step lines, ./log,
    every vort / psi / u / v / pres record and the find_min series (<= 1e-5), and the oracle at the first records.
    What this pins: that a user of the reference can switch (the boundary is complete for the reference's own callers), and that the
    oracle's restatement of main-shallow-water.cpp's loops agrees with the reference's compiled loops; the FFT on both sides is the
    engine's, so it is no FFTW pin."""
    import re
    import oracle_py as O
    import ref_numpy as R
    _build()
    n, dt = 768, 3.0
    runs = {}
    for tag in ("ref", "own"):
        d = tmp_path / tag
        (d / "input").mkdir(parents=True)
        (d / "output").mkdir()
        # "ref": the reference's own programs throughout; "own": product binaries ONLY -- host/makefield.out under the
        # reference's program name, host/vort_src_input.out, host/barotropic_main.out, host/invert_pres.out, host/find_min.out
        if tag == "own":
            os.symlink(os.path.join(HOST, "makefield.out"), str(d / "makefield-Kuo2004.out"))
        subprocess.check_call([os.path.join(REFDIR, "makefield-Kuo2004.out") if tag == "ref" else str(d / "makefield-Kuo2004.out")], cwd=str(d),
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        os.mkfifo(str(d / "vort_src_fifo"))
        prod = subprocess.Popen("%s > vort_src_fifo" % os.path.join(REFDIR if tag == "ref" else HOST, "vort_src_input.out"), shell=True, cwd=str(d),
                                stderr=subprocess.DEVNULL)
        main = [os.path.join(LINKDIR, "main-shallow-water.out"), "-fvort_src_fifo"] if tag == "ref" else \
               [os.path.join(HOST, "barotropic_main.out"), "-fvort_src_fifo"]
        res = subprocess.run(main, cwd=str(d), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
        prod.wait(timeout=60)
        assert res.returncode == 0, res.stderr[-2000:]
        lines = []
        for ln in (d / "log").read_text().split():                                  # example.sh:15
            mm = re.match(r"(.*/)psi(.*?\.bin)", ln)
            if mm:
                lines.append("%s=>%spres%s" % (ln, mm.group(1), mm.group(2)))
        inv = [os.path.join(LINKDIR, "invert_pres.out")] if tag == "ref" else [os.path.join(HOST, "invert_pres.out")]
        subprocess.run(inv, cwd=str(d), input="\n".join(lines) + "\n", stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, text=True, check=True, timeout=600)
        pres = sorted((f for f in os.listdir(str(d / "output")) if f.startswith("pres_step")), key=lambda f: int(re.findall(r"\d+", f)[0]))
        fm = subprocess.run([os.path.join(REFDIR if tag == "ref" else HOST, "find_min.out")], cwd=str(d), input="".join("output/%s\n" % f for f in pres),
                            stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, check=True)
        runs[tag] = (res.stdout, (d / "log").read_text(), fm.stdout, len(pres))
    steps_of = lambda out: [ln for ln in out.splitlines() if ln.startswith("# Step")]
    assert (tmp_path / "ref" / "input" / "initial_vorticity.bin").read_bytes() == (tmp_path / "own" / "input" / "initial_vorticity.bin").read_bytes()
    assert steps_of(runs["ref"][0]) == steps_of(runs["own"][0]) and len(steps_of(runs["ref"][0])) == 1200
    # the reference's plain driver main.cpp (no source path, rk4_c / dvortdt_c buffers: main.cpp:286-317), unmodified, on the same input
    d = tmp_path / "ref_main"
    (d / "input").mkdir(parents=True)
    (d / "output").mkdir()
    subprocess.check_call([os.path.join(REFDIR, "makefield-Kuo2004.out")], cwd=str(d), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    res = subprocess.run([os.path.join(LINKDIR, "main.out")], cwd=str(d), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    assert steps_of(res.stdout) == steps_of(runs["own"][0]) and (d / "log").read_text() == runs["own"][1]
    for step in (0, 500, 1100):
        for name in ("vort", "psi", "u", "v"):
            a = np.fromfile(str(d / "output" / ("%s_step_%d.bin" % (name, step))), dtype="<f4")
            b = np.fromfile(str(tmp_path / "own" / "output" / ("%s_step_%d.bin" % (name, step))), dtype="<f4")
            assert R.rel_l2(b, a) < 1e-5, ("main.cpp", name, step)
    assert runs["ref"][1] == runs["own"][1] and runs["ref"][3] == 12
    rd = lambda tag, f: np.fromfile(str(tmp_path / tag / "output" / f), dtype="<f4")
    worst = 0.0
    for step in range(0, 1200, 100):
        for name in ("vort", "psi", "u", "v", "pres"):
            a, b = rd("ref", "%s_step_%d.bin" % (name, step)), rd("own", "%s_step_%d.bin" % (name, step))
            assert a.size == n * n
            e = R.rel_l2(b, a)
            worst = max(worst, e)
            assert e < 1e-5, (name, step, e)
        assert np.array_equal(rd("ref", "vort_src_input_step_%d.bin" % step), rd("own", "vort_src_input_step_%d.bin" % step)), step
    print("reference pipeline on the engine vs the engine's own driver: worst record rel L2 %.2e" % worst)
    fa = np.array([[float(x) for x in ln.split()] for ln in runs["ref"][2].splitlines() if ln.strip()])
    fb = np.array([[float(x) for x in ln.split()] for ln in runs["own"][2].splitlines() if ln.strip()])
    # "x y minimum" per line, the 30 lowest values of every file (find_min.cpp:67-101); which of two nearly equal minima comes first
    # depends on the last bits, so the values are compared as sorted sets per file
    assert fa.shape == fb.shape and fa.shape[0] == 12 * 30
    for k in range(12):
        assert np.allclose(np.sort(fa[30 * k:30 * k + 30, 2]), np.sort(fb[30 * k:30 * k + 30, 2]), rtol=1e-4, atol=0.0), k
    # the oracle's restatement of the loop against the reference's compiled loop (FFT apart): the first two records
    m = O.Model(n, n, dt=dt)
    m.set_vort(np.fromfile(str(tmp_path / "ref" / "input" / "initial_vorticity.bin"), dtype="<f4").reshape(n, n))     # (the producer's source starts at step 2400)
    m.step(100)
    assert R.rel_l2(rd("ref", "vort_step_100.bin"), m.vort().ravel()) < 1e-5
