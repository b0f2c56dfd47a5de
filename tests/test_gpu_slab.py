"""GPU tests of the engine-driven multi-GPU step (fb_slab_*, csrc/fb_slab_driver.h) on the ONE GPU of the box.

All `world` ranks live in this process, one host thread each, on the same device; the all-to-all transposes go through the
in-process transport (fb_slab_connect_local: device-to-device copies ordered by the same kind of events the RCCL path
relies on), so what runs is the real schedule -- two streams per rank, field groups and row chunks pipelined, frozen
columns exchanged once -- with ranks racing each other.  The slab path must reproduce the fused single-GPU path bit for
bit wherever both use the same kernels.  BASELINE configs 4 and 5 run here at FULL size.
RCCL itself needs one GPU per rank; its call path is exercised with world = 1 (own block routed through ncclSend/ncclRecv)."""
import os
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _slab():
    from importlib import import_module
    return import_module("xlab-fftbarotropic_amd.slab")


def run_ranks(world, fn):
    """fn(rank) on `world` threads; re-raises the first failure."""
    errs, out = [None] * world, [None] * world

    def work(r):
        try:
            out[r] = fn(r)
        except BaseException as e:                      # noqa: BLE001 -- reported below
            errs[r] = e
    ts = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for e in errs:
        if e is not None:
            raise e
    return out


def slab_run(n, world, steps, v0, dt, src=None, ny=None, env=None):
    """vort after `steps` steps of the world-rank model (rows concatenated), plus the ranks' plan."""
    S = _slab()
    ny = ny or n
    hub = S.local_hub(world) if world > 1 else None
    plan = {}

    def rank_fn(r):
        m = S.EngineSlab(n, ny, dt=dt, rank=r, world=world, transport=hub)
        try:
            plan[r] = (m.field_groups, m.row_chunks, m.KA, m.KF)
            m.set_vort_local(S.local_rows(v0, r, world))
            if src is not None:
                m.set_source_local(S.local_rows(src, r, world))
            back = m.vort_local().cpu().numpy()
            m.step(steps)
            return back, m.vort_local().cpu().numpy()
        finally:
            m.close()
    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        res = run_ranks(world, rank_fn)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        if hub is not None:
            S.local_hub_destroy(hub)
    return np.concatenate([r[0] for r in res], axis=0), np.concatenate([r[1] for r in res], axis=0), plan[0]


@pytest.mark.parametrize("world,n,steps,env", [
    (1, 256, 3, None), (2, 256, 3, None), (4, 256, 2, None), (8, 512, 1, None), (2, 1024, 2, None), (4, 768, 2, None),
    (4, 1024, 2, {"FB_SLAB_FIELD_GROUPS": "4", "FB_SLAB_ROW_CHUNKS": "4"}),       # finest pipelining on a small grid
    (2, 512, 2, {"FB_SLAB_FIELD_GROUPS": "2", "FB_SLAB_ROW_CHUNKS": "8"}),
    (2, 512, 3, {"FB_SLAB_COL_GROUPS": "2"}),                                     # the stage pipelined by column groups (configs 4 and 5 take this path)
    (4, 1024, 2, {"FB_SLAB_COL_GROUPS": "2", "FB_SLAB_ROW_CHUNKS": "4"}),
    (8, 512, 2, {"FB_SLAB_COL_GROUPS": "2", "FB_SLAB_ROW_CHUNKS": "2"}),
])
def test_engine_slab_matches_fused_path(world, n, steps, env):
    import xlab_fftbarotropic_amd as X
    dt = 3.0
    v0 = X.make_field("elliptic", n)
    src = X.make_source_kuo2004(n)
    ref = X.Model(n, n, dt=dt)
    ref.set_vort(v0)
    ref.set_source(src)
    back_want = ref.vort().cpu().numpy()
    ref.step(steps)
    want = ref.vort().cpu().numpy()
    back, got, plan = slab_run(n, world, steps, v0, dt, src=src, env=env)
    if env and "FB_SLAB_FIELD_GROUPS" in env:
        assert plan[0] == int(env["FB_SLAB_FIELD_GROUPS"]) and plan[1] <= int(env["FB_SLAB_ROW_CHUNKS"])
    if env and "FB_SLAB_COL_GROUPS" in env:
        assert plan[0] == 1 and plan[1] <= int(env.get("FB_SLAB_ROW_CHUNKS", "1"))
    assert np.array_equal(back.view(np.uint32), back_want.view(np.uint32))            # set_vort / get_vort across the transposes
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("nx,world", [(512, 2), (128, 8)])
def test_engine_slab_4096_uses_the_headline_row_kernel(nx, world):
    """ny = 4096 on 2 and 8 ranks (different slab widths, both column groups present): the slab row pass is k_rowq too (slab-blocked addressing); the column
    side is the three-kernel path on both sides (nx < 4096 for the single-GPU reference) -> bit-identical."""
    import xlab_fftbarotropic_amd as X
    ny, steps = 4096, 2
    rng = np.random.default_rng(5)
    v0 = (rng.standard_normal((nx, ny)) * 1e-4).astype(np.float32)
    src = (rng.standard_normal((nx, ny)) * 1e-9).astype(np.float32)
    _, got, plan = slab_run(nx, world, steps, v0, 0.75, src=src, ny=ny)
    assert plan[2:] == _slab().slab_geometry(nx, ny, world)[1:] and plan[3] > 0     # active and frozen slabs both present
    ref = X.Model(nx, ny, dt=0.75)                      # three-kernel column path, k_rowq row pass
    ref.set_vort(v0)
    ref.set_source(src)
    ref.step(steps)
    assert np.array_equal(got.view(np.uint32), ref.vort().cpu().numpy().view(np.uint32))


def _invariants(v0, v1, with_source=False):
    assert np.isfinite(v1).all()
    m0, m1 = float(v0.astype(np.float64).mean()), float(v1.astype(np.float64).mean())
    if not with_source:
        assert abs(m1 - m0) <= 1e-6 * abs(m0) + 1e-12                               # (0,0) mode: exact in spectral space, fp32 sum here
        e0, e1 = float((v0.astype(np.float64) ** 2).sum()), float((v1.astype(np.float64) ** 2).sum())
        assert e1 <= e0 * (1 + 1e-6)                                                  # enstrophy does not grow


def test_config4_8192_gaussian_on_4_ranks_full_size():
    """BASELINE configs[3]: 8192 x 8192 gaussian vortex, dt = 3*1024/8192 s, ky slabs over 4 ranks -- full size.
    One RK4 step: the 4-rank engine path against the single-GPU run of the same kernels (bit for bit), the default single-GPU path, the oracle
    (main-shallow-water.cpp:277-338 == main.cpp:259-317 for a zero source) and the invariants."""
    import oracle_py as O
    import ref_numpy as R
    import xlab_fftbarotropic_amd as X
    n, world, dt = 8192, 4, 3.0 * 1024 / 8192
    v0 = X.make_field("gaussian", n)
    assert np.array_equal(v0[::64, ::64], O.make_field("gaussian", n)[::64, ::64])
    ref = X.Model(n, n, dt=dt)                                                        # default single-GPU path: k_rowh2 + k_col_full
    ref.set_vort(v0)
    s0 = ref.spectrum()[0, 0].item()
    ref.step(1)
    assert ref.spectrum()[0, 0].item() == s0                                          # mean vorticity conserved exactly
    want_default = ref.vort().cpu().numpy()
    del ref
    os.environ["FB_FULL_PASS"] = "0"                                                  # the three-kernel x pass: the kernels the ranks run
    try:
        ref = X.Model(n, n, dt=dt)
    finally:
        os.environ.pop("FB_FULL_PASS", None)
    ref.set_vort(v0)
    ref.step(1)
    want = ref.vort().cpu().numpy()
    del ref
    back, got, plan = slab_run(n, world, 1, v0, dt)
    assert plan == (1, 1, 976, 64)                                                    # pipelined by the two column groups (496 + 480 columns), rows not (84 us in all)
    assert R.rel_l2(back, v0) < 1e-6
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert R.rel_l2(got, want_default) < 2e-6                                         # same maths, another factorisation of the x transform
    _invariants(v0, got)
    mo = O.Model(n, n, dt=dt)
    mo.set_vort(v0)
    mo.step(1)
    assert R.rel_l2(got, mo.vort()) < 1e-5                                            # north-star bar, stated for 1000 steps
    assert R.rel_l2(want_default, mo.vort()) < 1e-5


def test_config5_16384_source_forced_on_8_ranks_full_size():
    """BASELINE configs[4]: 16384 x 16384, the source-forced variant (main-shallow-water.cpp:277-338: Kuo2004 initial field
    + the FIFO producer's cake of vort_src_input.cpp:35-46 switched on), dt = 3*1024/16384 s, ky slabs over 8 ranks --
    full size (about 25 GB of HBM with the single-GPU reference model alongside).  One RK4 step: 8-rank engine path
    against the fused single-GPU path bit for bit; invariants; the source really acts."""
    import ref_numpy as R
    import xlab_fftbarotropic_amd as X
    n, world, dt = 16384, 8, 3.0 * 1024 / 16384
    v0 = X.make_field("kuo2004", n)
    src = X.make_source_kuo2004(n)
    assert float(src.max()) > 0
    ref = X.Model(n, n, dt=dt)
    ref.set_vort(v0)
    s0 = ref.spectrum()[0, 0].item()
    ref.step(1)
    assert ref.spectrum()[0, 0].item() == s0
    unforced = ref.vort().cpu().numpy()
    _invariants(v0, unforced)
    ref.set_vort(v0)
    ref.set_source(src)
    ref.step(1)
    want = ref.vort().cpu().numpy()
    del ref
    assert not np.array_equal(want, unforced)
    gain = (want.astype(np.float64) - unforced.astype(np.float64)).sum() / (src.astype(np.float64).sum() * dt)
    assert abs(gain - 1.0) < 1e-3                                                     # d(mean vort)/dt = mean source
    back, got, plan = slab_run(n, world, 1, v0, dt, src=src)
    assert plan == (1, 2, 976, 64)                                                    # two column groups, two row chunks
    assert R.rel_l2(back, v0) < 1e-6
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    _invariants(v0, got, with_source=True)


@pytest.mark.parametrize("n,world,steps,env", [(512, 4, 40, {"FB_SLAB_FIELD_GROUPS": "4", "FB_SLAB_ROW_CHUNKS": "2"}),
                                               (512, 4, 40, {"FB_SLAB_COL_GROUPS": "2", "FB_SLAB_ROW_CHUNKS": "2"}),
                                               (4096, 4, 8, {"FB_SLAB_FIELD_GROUPS": "4", "FB_SLAB_ROW_CHUNKS": "4"}),
                                               (4096, 8, 6, None),
                                               (4096, 8, 6, {"FB_SLAB_TWO_STREAMS": "1"}),
                                               (512, 4, 40, None)])
def test_engine_slab_many_steps_stay_in_step(n, world, steps, env):
    """Ranks racing through many stages (thousands of event hand-offs between up to 16 streams), on a small grid where the host
    runs far ahead of the device and at the headline grid where kernels take long enough for every overlap to happen: any missing
    dependency between a rank's compute and communication streams shows up as a bit difference.  Plans with one field group and one
    row chunk (4096^2 on 8 ranks, 512^2 on 4) run on ONE stream per rank, the exchanges in line with the kernels;
    FB_SLAB_TWO_STREAMS=1 keeps the separate communication stream for the same plan."""
    import xlab_fftbarotropic_amd as X
    v0 = X.make_field("kuo2004", n)
    dt = 3.0 if n <= 1024 else 0.75
    os.environ["FB_FULL_PASS"] = "0"                    # the three-kernel x pass: the kernels the ranks run
    try:
        ref = X.Model(n, n, dt=dt)
    finally:
        os.environ.pop("FB_FULL_PASS", None)
    ref.set_vort(v0)
    ref.step(steps)
    _, got, _ = slab_run(n, world, steps, v0, dt, env=env)
    assert np.array_equal(got.view(np.uint32), ref.vort().cpu().numpy().view(np.uint32))


def test_transport_selftests():
    """fb_slab_transport_selftest through the in-process hub (4 ranks) and through RCCL with world = 1 and the own block
    routed through grouped ncclSend/ncclRecv (FB_RCCL_SELF=1): library load, ncclCommInitRank, the grouped call path."""
    import ctypes as C
    import xlab_fftbarotropic_amd as X
    S = _slab()
    L = X.lib()
    hub = S.local_hub(4)

    def fn(r):
        m = S.EngineSlab(256, rank=r, world=4, transport=hub)
        bad = C.c_size_t(123)
        X.package.binding.check(L.fb_slab_transport_selftest(m._h, 100003, C.byref(bad)))
        m.close()
        return bad.value
    assert run_ranks(4, fn) == [0, 0, 0, 0]
    S.local_hub_destroy(hub)
    os.environ["FB_RCCL_SELF"] = "1"
    try:
        m = S.EngineSlab(256, rank=0, world=1)
        idbuf = C.create_string_buffer(128)
        X.package.binding.check(L.fb_slab_unique_id(idbuf))
        X.package.binding.check(L.fb_slab_connect_rccl(m._h, idbuf))
        bad = C.c_size_t(123)
        X.package.binding.check(L.fb_slab_transport_selftest(m._h, 1 << 20, C.byref(bad)))
        assert bad.value == 0
        # what bench.py prints as rccl_ranks / devices comes from here: ncclCommCount, ncclCommUserRank and ncclCommCuDevice as the REAL
        # communicator answers them (the three symbols are dlsym'd from whichever RCCL the process has loaded; -1 would mean "not found")
        import torch
        info = m.transport_info()
        assert info["name"] == "rccl" and info["comm_ranks"] == 1 and info["comm_rank"] == 0, info
        assert info["hip_device"] == torch.cuda.current_device() and info["comm_device"] == info["hip_device"], info
        m.close()
        plain = S.EngineSlab(256, rank=0, world=1)                  # nothing connected: no communicator to ask
        i0 = plain.transport_info()
        assert i0["name"] == "none" and i0["comm_ranks"] == -1 and i0["hip_device"] == torch.cuda.current_device(), i0
        plain.close()
    finally:
        os.environ.pop("FB_RCCL_SELF", None)


def test_slab_model_world1_api():
    """SlabModel with world == 1 (no transport) equals Model."""
    import xlab_fftbarotropic_amd as X
    S = _slab()
    n = 256
    v0 = X.make_field("gaussian", n)
    m = S.SlabModel(n, n, rank=0, world=1)
    m.set_vort_local(v0)
    m.step(4)
    ref = X.Model(n, n)
    ref.set_vort(v0)
    ref.step(4)
    assert np.array_equal(m.vort_local().cpu().numpy().view(np.uint32), ref.vort().cpu().numpy().view(np.uint32))
    m.close()


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _bench_line_checks(d, also=True):
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["parallelism"] == "slab2"
    # what the line must carry for N > 1: the device ordinals, the model's prediction beside the value, rank 0's
    # single-GPU rate on the same grid from the same job, a spin-up, and the further grids as configs_run
    assert d["devices"] == [0, 0] and d["rccl_ranks"] is None                    # gloo rehearsal: no RCCL communicator to count
    assert d["predicted"]["steps_per_s"] > 0 and d["predicted"]["local_from"].startswith("measured")
    assert d["one_gpu_same_grid_steps_per_s"] > 0 and d["vs_1gpu_same_grid"] > 0 and d["spinup_steps"] >= 2
    assert d["local_passes_ms_per_step"] > 0 and d["config"]["slab"]["transport_selftest"] == "ok"
    if not also:
        assert d["configs_run"] == []
        return
    (c,) = d["configs_run"]
    assert c["grid"] == [256, 256] and c["value"] > 0 and c["one_gpu_same_grid_steps_per_s"] > 0 and c["predicted"]["steps_per_s"] > 0


def test_bench_starts_its_own_ranks_from_a_plain_shell(tmp_path):
    """`python bench.py --gpus 2 ...` with no launcher around it (the form of the driver's BENCH command): the parent starts the
    ranks as a child torch.distributed.run job before it has touched a GPU, relays rank 0's line and returns the child's code.  The child
    IS the contract's documented launch (python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 ...: one EngineSlab per
    process, the callback transport over gloo, 2 ranks sharing the GPU -- RCCL itself needs two devices), so this covers both forms."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--grid", "512", "--steps", "3", "--warmup", "1",
           "--cpu-steps", "0", "--also-grid", "256"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, env=env, cwd=str(tmp_path))
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{")                         # stdout is the line and nothing else
    _bench_line_checks(json.loads(lines[0]))


def test_engine_slab_two_processes_gloo_matches_single(tmp_path):
    """EngineSlab on 2 processes (one GPU, gloo callback transport) reproduces the single-process field bit for bit."""
    import subprocess
    import sys
    script = tmp_path / "w.py"
    script.write_text(
        "import os, sys, numpy as np, torch, torch.distributed as dist\n"
        "sys.path.insert(0, %r)\n"
        "from importlib import import_module\n"
        "import xlab_fftbarotropic_amd as X\n"
        "slab = import_module('xlab-fftbarotropic_amd.slab')\n"
        "torch.cuda.set_device(0); dist.init_process_group('gloo')\n"
        "r, w = dist.get_rank(), dist.get_world_size(); n = 256\n"
        "v0 = X.make_field('elliptic', n)\n"
        "m = slab.SlabModel(n, n, rank=r, world=w); m.set_vort_local(slab.local_rows(v0, r, w)); m.step(3)\n"
        "np.save(os.path.join(%r, 'rows%%d.npy' %% r), m.vort_local().cpu().numpy())\n"
        "m.close(); dist.destroy_process_group()\n" % (ROOT, str(tmp_path)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script)]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    import xlab_fftbarotropic_amd as X
    n = 256
    ref = X.Model(n, n)
    ref.set_vort(X.make_field("elliptic", n))
    ref.step(3)
    want = ref.vort().cpu().numpy()
    got = np.concatenate([np.load(str(tmp_path / ("rows%d.npy" % r))) for r in range(2)], axis=0)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("env", [{}, {"FB_SLAB_FIELD_GROUPS": "4", "FB_SLAB_ROW_CHUNKS": "2"}])
def test_engine_slab_rccl_on_several_gpus(tmp_path, env):
    """The product transport on real hardware: one process per GPU, ncclCommInitRank + grouped ncclSend/ncclRecv over xGMI, against
    the single-GPU run of the same kernels (bit for bit).  Needs at least two GPUs in one box -- skipped on the one-GPU boxes the
    builder has (there the RCCL call path is covered with world = 1 by test_transport_selftests and the schedule by the
    threads-as-ranks tests above).  Second case: the two-stream pipelined plan (field groups and row chunks forced)."""
    import subprocess
    import sys
    import torch
    ngpu = torch.cuda.device_count()
    if ngpu < 2:
        pytest.skip("needs >= 2 GPUs (RCCL refuses two ranks on one device)")
    world = 4 if ngpu >= 4 else 2
    script = tmp_path / "w.py"
    script.write_text(
        "import os, sys, numpy as np, torch, torch.distributed as dist\n"
        "sys.path.insert(0, %r)\n"
        "from importlib import import_module\n"
        "import xlab_fftbarotropic_amd as X\n"
        "slab = import_module('xlab-fftbarotropic_amd.slab')\n"
        "r = int(os.environ['RANK']); torch.cuda.set_device(int(os.environ['LOCAL_RANK']))\n"
        "dist.init_process_group('nccl', device_id=torch.device('cuda', torch.cuda.current_device()))\n"
        "w = dist.get_world_size(); n = 1024\n"
        "v0 = X.make_field('elliptic', n); src = X.make_source_kuo2004(n)\n"
        "m = slab.SlabModel(n, n, rank=r, world=w); assert m.transport.startswith('rccl'), m.transport\n"
        "m.set_vort_local(slab.local_rows(v0, r, w)); m.set_source_local(slab.local_rows(src, r, w)); m.step(5)\n"
        "np.save(os.path.join(%r, 'rows%%d.npy' %% r), m.vort_local().cpu().numpy())\n"
        "m.close(); dist.destroy_process_group()\n" % (ROOT, str(tmp_path)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script)]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=dict(os.environ, **env))
    assert res.returncode == 0, res.stderr[-3000:]
    import xlab_fftbarotropic_amd as X
    n = 1024
    ref = X.Model(n, n)
    ref.set_vort(X.make_field("elliptic", n))
    ref.set_source(X.make_source_kuo2004(n))
    ref.step(5)
    got = np.concatenate([np.load(str(tmp_path / ("rows%d.npy" % r))) for r in range(world)], axis=0)
    assert np.array_equal(got.view(np.uint32), ref.vort().cpu().numpy().view(np.uint32))


@pytest.mark.parametrize("env", [{}, {"FB_SLAB_FIELD_GROUPS": "4", "FB_SLAB_ROW_CHUNKS": "2"}])
def test_cpp_driver_one_process_per_gpu_rccl(tmp_path, env):
    """host/barotropic_main.out --world P --rank r --comm-file F --launch-token T: the C++ multi-process flow over RCCL
    (rank 0 publishes the ncclUniqueId through the comm file, a stale record of another launch lies at the path and must be
    ignored), once with the default plan and once with the two-stream pipelined one, against the single-GPU run of
    the same configuration.  Needs >= 2 GPUs in one box."""
    import subprocess
    import torch
    import oracle_py as O
    ngpu = torch.cuda.device_count()
    if ngpu < 2:
        pytest.skip("needs >= 2 GPUs (RCCL refuses two ranks on one device)")
    world = 4 if ngpu >= 4 else 2
    host = os.path.join(ROOT, "xlab-fftbarotropic_amd", "host")
    subprocess.check_call(["make", "-s", "-C", host])
    exe = os.path.join(host, "barotropic_main.out")
    n, steps = 1024, 21
    outs = {}
    for tag in ("one", "many"):
        d = tmp_path / tag
        (d / "input").mkdir(parents=True)
        (d / "output").mkdir()
        O.make_field("elliptic", n).tofile(str(d / "input" / "initial_vorticity.bin"))
        base = [exe, "--npts", str(n), "--steps", str(steps), "--record-step", "10"]
        if tag == "one":
            res = [subprocess.run(base, cwd=str(d), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)]
        else:
            comm = str(d / "comm")
            subprocess.check_call([os.path.join(host, "comm_bootstrap_check.out"), "publish", comm, "an-older-launch", "7"])
            procs = [subprocess.Popen(base + ["--world", str(world), "--rank", str(r), "--comm-file", comm, "--launch-token", "t-%d" % os.getpid(),
                                              "--comm-timeout", "120"], cwd=str(d), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                                      env=dict(os.environ, **env)) for r in range(world)]
            res = []
            for p in procs:
                try:
                    so, se = p.communicate(timeout=600)
                except subprocess.TimeoutExpired:
                    for q in procs:
                        q.kill()
                    raise
                res.append(subprocess.CompletedProcess(p.args, p.returncode, so, se))
        for r in res:
            assert r.returncode == 0, r.stderr[-2000:]
        outs[tag] = res[0].stdout
    assert outs["one"] == outs["many"]                                        # rank 0 owns the banner and the step lines
    rd = lambda tag, f: np.fromfile(str(tmp_path / tag / "output" / f), dtype="<f4")
    import ref_numpy as R
    for step in (0, 10, 20):
        a, b = rd("one", "vort_step_%d.bin" % step), rd("many", "vort_step_%d.bin" % step)
        assert a.size == n * n and R.rel_l2(b, a) < 2e-6, step                 # single-pass x transform on one GPU, three kernels on the ranks
        for name in ("psi", "u", "v"):
            assert R.rel_l2(rd("many", "%s_step_%d.bin" % (name, step)), rd("one", "%s_step_%d.bin" % (name, step))) < 2e-6
