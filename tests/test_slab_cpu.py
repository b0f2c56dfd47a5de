"""CPU (gloo, world_size 2 and 4) tests of the slab-decomposed driver's exchange logic.

The HIP kernels cannot run here; the compute backend is the fp64 numpy test double
(tests/slab_numpy_backend.py) that obeys the same buffer-layout contract, so what is under test
is xlab-fftbarotropic_amd/slab.py: buffer geometry, who sends what to whom, phase order."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, steps, with_src, out_path, env=None):
    os.environ.update(env or {})
    for p in (ROOT, HERE, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from importlib import import_module
    import ref_numpy as R
    from slab_numpy_backend import NumpyBackend
    slab = import_module("xlab-fftbarotropic_amd.slab")
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        import oracle_py as O
        v0 = O.make_field("elliptic", n).astype(np.float64)
        src = None
        if with_src:
            s = np.zeros((n, n), dtype=np.float32)
            O.add_cake(s, 6e5, 6e5, 6e5 / 2 + 5e4, 6e5 / 2, 3e-3 / 10800.0, 3e4)
            src = s.astype(np.float64)
        be = NumpyBackend(n, n, 6e5, 6e5, 6.5, 3.0, rank, world)
        m = slab.SlabModel(n, n, rank=rank, world=world, backend=be, dist=dist)
        m.set_vort_local(slab.local_rows(v0, rank, world))
        if src is not None:
            m.set_source_local(slab.local_rows(src, rank, world))
        back0 = m.vort_local().numpy().copy()
        m.step(steps)
        rows = m.vort_local().numpy()
        # single-process fp64 reference of the same maths
        ref = R.Model64(n, n)
        ref.set_vort(v0)
        if src is not None:
            ref.src = src
        ref.step(steps)
        want = slab.local_rows(ref.vort(), rank, world)
        err0 = float(np.abs(back0 - slab.local_rows(v0, rank, world)).max())
        err = R.rel_l2(rows, want)
        np.save(out_path % rank, np.array([err0, err]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,steps,with_src,env", [(2, 64, 3, False, None), (2, 64, 2, True, None), (4, 64, 2, False, None), (2, 256, 1, True, None),
                                                          (2, 256, 2, True, {"FB_SLAB_COL_GROUPS": "2"}),      # the stage pipelined by column groups
                                                          (4, 128, 2, False, {"FB_SLAB_COL_GROUPS": "2"})])
def test_slab_exchange_matches_single_process(tmp_path, world, n, steps, with_src, env):
    port = _free_port()
    out = str(tmp_path / "err_%d.npy")
    mp.spawn(_worker, args=(world, port, n, steps, with_src, out, env), nprocs=world, join=True)
    for r in range(world):
        err0, err = np.load(out % r)
        assert err0 < 1e-12, "c2r(r2c(x)) across the transposes, rank %d" % r
        assert err < 1e-10, "rank %d: rel L2 %g" % (r, err)


def test_slab_geometry_plan_and_world1():
    """Host logic of the decomposition, no GPU: slab.py's geometry and per-stage schedule against the engine's own
    (fb_slab_geometry / fb_slab_plan in libfftbaro.so), the even split of the ACTIVE columns, and world == 1."""
    import ctypes as C
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    from importlib import import_module
    import ref_numpy as R
    import xlab_fftbarotropic_amd as X
    from slab_numpy_backend import NumpyBackend
    slab = import_module("xlab-fftbarotropic_amd.slab")
    L = X.lib()
    # BASELINE configs 4 and 5, the headline grid on 2..8 ranks, a 3*2^k grid and a small one whose active slabs cover everything
    cases = [(8192, 8192, 4), (16384, 16384, 8), (4096, 4096, 1), (4096, 4096, 2), (4096, 4096, 4), (4096, 4096, 8), (768, 768, 4), (64, 64, 8), (256, 1024, 2)]
    for nx, ny, w in cases:
        xl, ka, kf = C.c_int(), C.c_int(), C.c_int()
        assert L.fb_slab_geometry(nx, ny, w, C.byref(xl), C.byref(ka), C.byref(kf)) == 0
        assert (xl.value, ka.value, kf.value) == slab.slab_geometry(nx, ny, w), (nx, ny, w)
        fg, ch, ops = slab.engine_plan(nx, ny, w)
        assert (fg, ch) == slab.stage_plan(nx, ny, w) and ops == slab.stage_schedule(nx, ny, w), (nx, ny, w)
        ng, cols = C.c_int(), (C.c_int * 2)()
        assert L.fb_slab_col_groups(nx, ny, w, C.byref(ng), cols) == 0
        groups = slab.slab_col_groups(nx, ny, w)
        assert [cols[g] for g in range(ng.value)] == groups and sum(groups) == ka.value and all(g % 16 == 0 and g > 0 for g in groups), (nx, ny, w)
        hy = ny // 2 + 1
        assert w * ka.value + w * kf.value >= hy and ka.value % 16 == 0 and kf.value % 16 == 0
        # every column inside the dealiasing circle lies in an active slab, and the last rank is not idle:
        jmax = int(np.ceil(np.sqrt(2.0) * np.ceil(ny / 3.0))) if nx == ny else None
        if jmax is not None:
            assert w * ka.value >= jmax - 1
            if jmax >= 16 * w:                                                 # (slabs are whole 16-column tiles)
                assert (w - 1) * ka.value < jmax                               # active columns reach into the last rank's slab
    assert slab.slab_geometry(8192, 8192, 4) == (2048, 976, 64)
    assert slab.slab_geometry(16384, 16384, 8) == (2048, 976, 64)
    assert slab.slab_geometry(4096, 4096, 8) == (512, 256, 16)
    # configs 4 and 5 and the headline grid on two ranks are pipelined by column groups (a group's four fields leave together);
    # smaller slabs keep one group and cut the derivative exchange by fields where that pays
    assert slab.slab_col_groups(8192, 8192, 4) == [496, 480] and slab.slab_col_groups(16384, 16384, 8) == [496, 480]
    assert slab.slab_col_groups(4096, 4096, 2) == [496, 480] and slab.slab_col_groups(4096, 4096, 4) == [496] and slab.slab_col_groups(4096, 4096, 8) == [256]
    assert slab.stage_plan(8192, 8192, 4) == (1, 1) and slab.stage_plan(16384, 16384, 8) == (1, 2)
    assert slab.stage_plan(4096, 4096, 2) == (1, 1) and slab.stage_plan(4096, 4096, 4) == (1, 1) and slab.stage_plan(4096, 4096, 8) == (1, 1) and slab.stage_plan(4096, 4096, 1) == (1, 1)
    assert [k for k, _ in slab.stage_schedule(16384, 16384, 8)] == [3, 4, 3, 4, 5, 6, 2, 5, 6, 2]
    assert L.fb_slab_plan(1000, 1000, 2, None, None, None, 0) == 0             # unsupported grid
    n = 32
    rng = np.random.default_rng(0)
    v0 = 1e-3 * rng.standard_normal((n, n))
    be = NumpyBackend(n, n, 6e5, 6e5, 6.5, 3.0, 0, 1)
    m = slab.SlabModel(n, n, rank=0, world=1, backend=be)
    m.set_vort_local(v0)
    m.step(2)
    ref = R.Model64(n, n)
    ref.set_vort(v0)
    ref.step(2)
    assert R.rel_l2(m.vort_local().numpy(), ref.vort()) < 1e-10


def _connect_worker(rank, world, port, fail_rank, fail_step, out_path):
    """EngineSlab._connect_rccl over gloo with a stand-in for the engine library: `fail_step` ("id" on rank 0, "init" on
    `fail_rank`) reports an error; what is under test is that EVERY rank then raises, after the same collectives."""
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    from importlib import import_module
    slab = import_module("xlab-fftbarotropic_amd.slab")
    B = import_module("xlab-fftbarotropic_amd.binding")
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        seen = {}

        class FakeLib:
            def fb_slab_unique_id(self, buf):
                if fail_step == "id":
                    return 1
                buf.raw = bytes(range(128))
                return 0

            def fb_slab_connect_rccl(self, h, buf):
                seen["id"] = bytes(buf.raw[:128])
                return 1 if (fail_step == "init" and rank == fail_rank) else 0

        class FakeB:
            FftBaroError = B.FftBaroError

            @staticmethod
            def check(status):
                if status != 0:
                    raise B.FftBaroError("stand-in failure")

        class Fake:
            pass
        f = Fake()
        f.torch, f.dist, f.B, f.L, f.rank, f.world, f._h = torch, dist, FakeB, FakeLib(), rank, world, None
        f._agree = lambda ok, what: slab.EngineSlab._agree(f, ok, what)
        try:
            slab.EngineSlab._connect_rccl(f)
            outcome = "connected"
        except B.FftBaroError as e:
            outcome = "raised: %s" % e
        dist.barrier()                                             # the ranks are still in step: nobody is stuck in a stray collective
        ok_id = seen.get("id") == bytes(range(128)) if fail_step != "id" else "id" not in seen
        open(out_path % rank, "w").write("%s|%s" % (outcome, ok_id))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("fail_rank,fail_step", [(None, None), (0, "id"), (1, "init"), (0, "init")])
def test_rccl_connect_outcome_is_collective(tmp_path, fail_rank, fail_step):
    """a failed RCCL bootstrap must fail on ALL ranks -- no rank may fall back on its own."""
    world, port = 2, _free_port()
    out = str(tmp_path / "outcome_%d.txt")
    mp.spawn(_connect_worker, args=(world, port, fail_rank, fail_step, out), nprocs=world, join=True)
    got = [open(out % r).read().split("|") for r in range(world)]
    for r in range(world):
        assert got[r][1] == "True", got
        if fail_step is None:
            assert got[r][0] == "connected", got
        else:
            assert got[r][0].startswith("raised"), got


def test_predicted_step_model_reproduces_the_design_table():
    """slab.predicted_step_ms is DESIGN.md section 6's model as numbers (what bench.py prints as `predicted` beside a multi-GPU
    `value`): link time of 5 fields x XL x KA complex per peer at the assumed rate + the local work no transfer hides.  Host logic."""
    from importlib import import_module
    slab = import_module("xlab-fftbarotropic_amd.slab")
    # per peer and stage 80 MB at 4096^2 / 2, 8192^2 / 4 and 16384^2 / 8: 1.33 ms at 60 GB/s + 2 group latencies
    for n, world in ((4096, 2), (8192, 4), (16384, 8)):
        _, t = slab.predicted_step_ms(n, n, world, 1.0)
        xl, ka, _kf = slab.slab_geometry(n, n, world)
        assert abs(t["t_link_ms_per_stage"] - (5 * xl * ka * 8 / 60e9 * 1e3 + 0.06)) < 1e-9 and 1.3 < t["t_link_ms_per_stage"] < 1.5
    # the measured rank-local step times of DESIGN.md section 6 give the table's rates
    for n, world, local, lo, hi in ((4096, 2, 0.85, 150, 200), (4096, 4, 0.33, 450, 700), (4096, 8, 0.23, 1000, 1400),
                                    (8192, 4, 1.47, 150, 190), (16384, 8, 3.11, 150, 190)):
        ms, t = slab.predicted_step_ms(n, n, world, local)
        assert lo < 1e3 / ms < hi, (n, world, 1e3 / ms)
        assert t["exposed_ms_per_stage"] <= t["local_ms_per_stage"] + 1e-12 and t["local_from"].startswith("measured")
    # nothing pipelined (one field group, one row chunk, one column group): everything local is exposed
    _, t = slab.predicted_step_ms(4096, 4096, 8, 0.23)
    assert abs(t["exposed_ms_per_stage"] - t["local_ms_per_stage"]) < 1e-12
    ms1, t1 = slab.predicted_step_ms(4096, 4096, 1)
    assert t1["t_link_ms"] == 0.0 and ms1 > 0
    # without a measurement the local passes are priced at 22.4 C / world at 5 TB/s
    _, t = slab.predicted_step_ms(8192, 8192, 4)
    assert t["local_from"].startswith("22.4 C") and 0.2 < t["local_ms_per_stage"] < 0.4


def test_bench_parent_reports_a_failed_child_job(tmp_path):
    """`python bench.py --gpus 2` from a plain shell starts its ranks as a child job (bench.self_launch).  Here there is no GPU, so the
    ranks fail at start-up: the parent must come back with a non-zero code, an empty stdout (no half line) and the reason on stderr --
    not hang, and not exec anything."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    if torch.cuda.is_available():
        pytest.skip("GPU present: the GPU suite runs the working launch")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--grid", "256", "--steps", "1",
                          "--warmup", "0", "--cpu-steps", "0", "--launch-timeout", "240"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         timeout=400, env=env, cwd=str(tmp_path))
    assert res.returncode != 0 and res.stdout.strip() == ""
    assert "printed no line" in res.stderr
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "os.exec" not in src and "execv" not in src


_FAKE_RANK = r'''
import json, os, sys, time
open(os.path.join(%(dir)r, "pid.%%s" %% os.environ["RANK"]), "w").write(str(os.getpid()))
if os.environ["RANK"] == "0":
    with open(os.environ["FB_BENCH_LINE_FILE"], "w") as f:
        json.dump({"metric": "stand-in", "value": 1.0, "n_gpus": int(os.environ["WORLD_SIZE"])}, f)
time.sleep(600)                      # a job that hangs after its headline run
'''
_PARENT = r'''
import sys
sys.path.insert(0, %(root)r)
import bench
bench.__file__ = %(rank)r            # self_launch starts `torch.distributed.run <this file> <args>`
sys.argv = ["bench.py", "--gpus", "2", "--launch-timeout", %(timeout)r]
raise SystemExit(bench.self_launch(bench.parse_args(sys.argv[1:])))
'''


def _alive(pid):
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return False
    except PermissionError:
        return True
    try:                                                     # a zombie waiting to be reaped does not count as running
        return open("/proc/%d/stat" % pid).read().split(")")[-1].split()[0] != "Z"
    except OSError:
        return False


def _wait_for_pids(tmp_path, n, timeout=120):
    import time
    t_end = time.time() + timeout
    while time.time() < t_end:
        got = [p for p in (tmp_path / ("pid.%d" % r) for r in range(n)) if p.exists() and p.read_text().strip()]
        if len(got) == n:
            return [int(p.read_text()) for p in got]
        time.sleep(0.5)
    raise AssertionError("the stand-in ranks did not start")


def _launch_parent(tmp_path, timeout_s):
    import subprocess
    rank = tmp_path / "fake_rank.py"
    rank.write_text(_FAKE_RANK % {"dir": str(tmp_path)})
    parent = tmp_path / "parent.py"
    parent.write_text(_PARENT % {"root": ROOT, "rank": str(rank), "timeout": str(timeout_s)})
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.Popen([sys.executable, str(parent)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=str(tmp_path))


def test_bench_parent_ends_a_hung_job_and_still_prints_the_headline_line(tmp_path):
    """bench.self_launch with ranks that hang after their headline run (a stand-in rank script: no GPU needed): when --launch-timeout
    expires the parent ends the ranks' whole process group, prints the line rank 0 had left, marked `incomplete`, and returns non-zero;
    no rank is left behind."""
    import json
    import time
    p = _launch_parent(tmp_path, 20)
    pids = _wait_for_pids(tmp_path, 2)
    out, err = p.communicate(timeout=180)
    assert p.returncode != 0
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1, (out, err[-1500:])
    d = json.loads(lines[0])
    assert d["metric"] == "stand-in" and d["n_gpus"] == 2 and "launch timeout" in d["incomplete"]
    time.sleep(1.0)
    assert not any(_alive(pid) for pid in pids), pids


def test_bench_parent_takes_its_ranks_along_when_it_is_stopped(tmp_path):
    """The ranks run in a session of their own (so that exactly their process group can be ended), hence nobody but the parent ends them:
    a SIGTERM to the parent -- a harness time limit -- must not leave them on the node's GPUs.  Stand-in ranks, no GPU needed."""
    import json
    import signal
    import time
    p = _launch_parent(tmp_path, 500)
    pids = _wait_for_pids(tmp_path, 2)
    assert all(_alive(pid) for pid in pids)
    p.send_signal(signal.SIGTERM)
    out, err = p.communicate(timeout=120)
    assert p.returncode == 130, (p.returncode, err[-1500:])
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1 and "the parent was stopped" in json.loads(lines[0])["incomplete"]      # even then: the headline line
    time.sleep(1.0)
    assert not any(_alive(pid) for pid in pids), pids


def test_bench_ranks_do_not_outlive_a_killed_parent(tmp_path):
    """SIGKILL cannot be handled: the launcher is started with PR_SET_PDEATHSIG, so the kernel sends it SIGTERM when the parent dies and
    torch.distributed.run shuts its workers down.  Stand-in ranks, no GPU needed."""
    import signal
    import time
    p = _launch_parent(tmp_path, 500)
    pids = _wait_for_pids(tmp_path, 2)
    p.send_signal(signal.SIGKILL)
    p.communicate(timeout=60)
    t_end = time.time() + 90                                  # the elastic agent gives its workers a grace period
    while time.time() < t_end and any(_alive(pid) for pid in pids):
        time.sleep(1.0)
    assert not any(_alive(pid) for pid in pids), pids
