#!/usr/bin/env python3
"""bench.py -- RK4 steps/s of the pseudospectral barotropic-vorticity hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N>1: under torch.distributed.run --nproc-per-node N as the contract says, or from a plain shell -- then this process starts that job
  itself as a child, before touching a GPU, and passes rank 0's line on)

One "step" = one RK4 step (4 stages = 16 c2r + 4 r2c 2-D FFTs + all pointwise work of
main.cpp:286-317) of the synthetic workload below, state resident in HBM.  Prints ONE JSON line.

Workload at N=1: BASELINE.json configs[2] -- 4096x4096 Kuo2004 initial field, fp32,
dt = 3*1024/4096 s (the reference's dt=3 s is unstable above N~2300).
For N>1 the same grid is split into x-row / ky-column slabs (strong scaling; the engine drives the
RCCL all-to-all transposes itself, csrc/fb_slab_driver.h), see DESIGN.md section 6.  The N>1 line also carries: `rccl_ranks`
(ncclCommCount as every rank's communicator reports it), `devices` (HIP ordinals), `predicted` (the model of DESIGN.md section 6
beside the measured `value`), `one_gpu_same_grid_steps_per_s` / `vs_1gpu_same_grid` (rank 0 alone on the same grid in the same job)
and `configs_run`: at N = 4 also 8192^2 gaussian (BASELINE configs[3]), at N = 8 also 16384^2 source-forced (configs[4]), each with
its own single-GPU rate.  At N=1 a `driver` leg times the C++ drop-in driver itself (1000 steps, record_step 100).

Roofline conventions: `roofline.achieved` uses the contract's algorithmic bytes (SURVEY.md
8(d): 8 N^2 per 1-D pass over a field, 320 N^2 per step); next to it the line carries the bytes the kernels really
move (`traffic`, from the committed rocprofv3 PMC passes), `traffic_frac` = those bytes / launch time / peak, and at
step level `tight_frac` against the fully fused lower bound of 256 N^2.
"""
import argparse
import glob
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s measured achievable

# algorithmic bytes per launch, in units of N^2 bytes (every 1-D pass over a field
# reads + writes one N^2 float-equivalent array = 8 N^2 B; the x pass is two sub-pass kernels, each
# credited half).  Sum over the four kernels of a stage = 80 N^2; x 4 stages = 320 N^2 per step.
ALG_N2 = {"k_col_strided_bwd4": 16.0, "k_row_fused": 40.0, "k_col_strided_fwd1": 4.0, "k_col_mid": 20.0,
          "k_col_full": 40.0}      # the single-pass x transform does the work of the three column kernels
# kernel-name prefixes of the classes in the rocprofv3 output (profiles/*_pmc_traffic_<n>.json)
PMC_PREFIX = {"k_col_strided_bwd4": ("k_col_strided<", ", 1>"), "k_row_fused": ("k_row", ""), "k_col_strided_fwd1": ("k_col_strided<", ", -1>"),
              "k_col_mid": ("k_col_mid<", ""), "k_col_full": ("k_col_full", "")}


def pmc_profile(n):
    """The newest committed PMC summary for this grid: (path relative to the repo, parsed json) or (None, None)."""
    best = None
    for p in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic_%d.json" % n)):
        m = re.match(r"r(\d+)_([a-z]+)_pmc", os.path.basename(p))
        if m:
            key = (int(m.group(1)), len(m.group(2)), m.group(2))
            if best is None or key > best[0]:
                best = (key, p)
    if best is None:
        return None, None
    try:
        return os.path.relpath(best[1], ROOT), json.load(open(best[1]))
    except ValueError:
        return None, None


def pmc_traffic(prof, kernel, launches_per_step=4):
    """HBM-side bytes per launch of a kernel class (average over its template instances, weighted by launches)."""
    if not prof:
        return None
    pre, suf = PMC_PREFIX[kernel]
    tot = cnt = 0.0
    for k, d in prof.get("kernels", {}).items():
        if k.startswith(pre) and k.endswith(suf) if suf else k.startswith(pre):
            if kernel == "k_row_fused" and d.get("launches", 0) < 8:          # set_vort / get_vort launches of the non-fused modes
                continue
            if kernel == "k_col_full" and k.startswith("k_col_full<4"):       # the PRIME launch after set_vort is not a stage
                continue
            if kernel == "k_col_strided_bwd4":
                # per[kernel] is the duration of the four-field launch of an RK stage; the same kernel also runs single-field
                # launches in set_vort / get_vort: only the large launches count (tools/pmc_summary.py)
                if "hbm_bytes_per_big_launch_corrected" not in d:
                    return None                                               # an older summary that mixed both: no figure rather than a wrong one
                tot += d["hbm_bytes_per_big_launch_corrected"] * d["launches_big"]
                cnt += d["launches_big"]
                continue
            tot += d["hbm_bytes_per_launch_corrected"] * d["launches"]
            cnt += d["launches"]
    return tot / cnt if cnt else None


def oracle_source(O, n):
    """The FIFO producer's cake (vort_src_input.cpp:35-46) as the oracle builds it."""
    import numpy as np
    src = np.zeros((n, n), dtype=np.float32)
    O.add_cake(src, 600000.0, 600000.0, 600000.0 / 2 + 50000.0, 600000.0 / 2, 3e-3 / 10800.0, 30000.0)
    return src


def cpu_leg(n, dt, kind, steps, threads, source=False):
    import ctypes
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    try:
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(threads)
    except OSError:
        threads = 1
    m = O.Model(n, n, dt=dt)
    m.set_vort(O.make_field(kind, n))
    if source:
        m.set_source(oracle_source(O, n))
    m.step(1)
    t0 = time.perf_counter()
    m.step(steps)
    return steps / (time.perf_counter() - t0), threads


def cpu_baseline(n, dt, kind, steps, source=False):
    """The oracle ("port": the reference's unfused loop structure with its own FFT) on the GPU box's host cores: ONE core is
    the faithful figure -- the reference is single-threaded (Makefile:2, no threads anywhere) -- plus the same code on all
    cores (OpenMP) and the small configs, as SURVEY.md 8(d) asks.  Bounded: about 20-30 s in all."""
    nproc = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    share, how = nproc, "every core the process may run on"
    try:                                                # a container's CPU quota, not the host's core count, is what this process can use
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            share, how = max(1, -(-int(q) // int(per))), "the cgroup CPU quota"
    except (OSError, ValueError):
        pass
    if share == nproc and nproc > 16:
        share, how = 16, "the GPU box's CPU share for one GPU (16), nproc reports the whole host"
    v1, _ = cpu_leg(n, dt, kind, steps, 1, source)
    vall, used = cpu_leg(n, dt, kind, max(steps, 5), share, source)
    small = {}
    for ns, ks in ((256, 40), (1024, 20)):
        if ns < n:
            v, _ = cpu_leg(ns, 3.0, "elliptic", ks, 1)
            small["%dx%d elliptic, 1 core" % (ns, ns)] = v
    return {"value": v1, "unit": "steps/s", "cores": 1, "kind": "port",
            "sample": "%dx%d %s%s, %d RK4 steps after 1 warm-up, oracle/liboracle.so (own FFT, reference loop structure), 1 thread"
                      % (n, n, kind, " + source" if source else "", steps),
            "all_cores": {"value": vall, "unit": "steps/s", "cores": used, "nproc": nproc,
                          "sample": "same workload, %d steps, OpenMP over the oracle's loops and FFT batches; threads = %s" % (max(steps, 5), how)},
            "other_configs_steps_per_s": small}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--grid", "--n", dest="n", type=int, default=4096, help="grid points per side")
    ap.add_argument("--kind", default=None, help="initial field (default: the BASELINE config of the grid: 4096 kuo2004, 8192 gaussian, "
                                                  "16384 kuo2004 + source; smaller grids elliptic)")
    ap.add_argument("--source", type=int, default=None, help="1: the source-forced variant (main-shallow-water.cpp:277-338) with the FIFO "
                                                             "producer's cake switched on; default: on for --grid 16384 (BASELINE configs[4])")
    ap.add_argument("--cpu-steps", type=int, default=5, help="oracle steps for the 1-core cpu_baseline leg (0 = skip cpu_baseline)")
    ap.add_argument("--spinup-steps", type=int, default=None,
                    help="untimed device spin-up before the warm-up: this many steps, then the state is reset (default: ~30 ms worth; 0 = none)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--also-grid", default="auto",
                    help="N>1: further grids run in the same job as `configs_run` (comma separated; 'auto' = 8192 at N=4 -- BASELINE configs[3] -- "
                         "and 16384 at N=8 -- configs[4]; 'none' = only --grid)")
    ap.add_argument("--driver-steps", type=int, default=None,
                    help="N=1: the `driver` leg -- the C++ drop-in driver (host/barotropic_main.out) on the same workload for this many steps with "
                         "record_step 100 (configuration.hpp:34-36), records into a temporary directory (default: 1000 at the default grid, else 0 = off)")
    ap.add_argument("--launch-timeout", type=float, default=540.0,
                    help="N>1 started from a plain shell: seconds before the child job (its whole process group) is ended; the headline line rank 0 has "
                         "left by then is still printed, marked incomplete (an 8-rank job with its two configs takes about 1.5 minutes)")
    return ap.parse_args(argv)


def self_launch(args):
    """`python bench.py --gpus N` from a plain shell (N > 1, no WORLD_SIZE): start the ranks as a CHILD job -- torch.distributed.run,
    one process per GPU -- before this process has touched a GPU, pass rank 0's JSON line on and return the child's exit code.
    (Never exec: a process image must not be replaced on these hosts once a GPU is initialised, and this one stays to relay.)"""
    import signal
    import socket
    import subprocess
    import tempfile
    import threading
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    fd, line_file = tempfile.mkstemp(prefix="bench_line_", suffix=".json")
    os.close(fd)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: what RCCL needs across processes on these hosts
    env["FB_BENCH_LINE_FILE"] = line_file                      # rank 0 leaves the line here as soon as the headline run is done
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    # The ranks live in a session of their own so that exactly THEIR process group can be ended -- which also means nobody else ends them:
    # if this parent is told to stop (a harness time limit, Ctrl-C) it must take them along, or they stay on the node's GPUs.
    def die_with_parent():                                      # in the launcher, between fork and exec: PR_SET_PDEATHSIG = 1
        import ctypes
        ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, int(signal.SIGTERM))
    # (a SIGKILL to this parent cannot be caught here; the kernel then tells the launcher, which shuts its workers down)
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True, start_new_session=True, preexec_fn=die_with_parent)
    state = {"why": None}

    def end_child(why):
        if state["why"] is None:
            state["why"] = why
        for sig, grace in ((signal.SIGTERM, 10.0), (signal.SIGKILL, 0.0)):
            try:
                os.killpg(child.pid, sig)                       # the exact process group started above
            except (ProcessLookupError, PermissionError):
                return
            t_end = time.time() + grace
            while time.time() < t_end:
                if child.poll() is not None:
                    break
                time.sleep(0.2)

    def on_signal(signum, frame):
        raise KeyboardInterrupt("signal %d" % signum)
    old_handlers = {sg: signal.signal(sg, on_signal) for sg in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP)}
    timer = threading.Timer(args.launch_timeout, end_child, args=("launch timeout of %g s" % args.launch_timeout,))
    timer.daemon = True
    timer.start()
    line, rc = None, None
    try:
        for ln in child.stdout:
            t = ln.strip()
            if t.startswith("{") and t.endswith("}"):
                try:
                    json.loads(t)
                    line = t
                    continue
                except ValueError:
                    pass
            sys.stderr.write(ln)
        rc = child.wait()
    except KeyboardInterrupt as e:                              # this parent was told to stop: the ranks go first
        end_child("the parent was stopped (%s)" % e)
        rc = child.wait()
    finally:
        timer.cancel()
        if child.poll() is None:                                # whatever brought us here, no rank outlives the parent
            end_child("parent leaving")
            child.wait()
        for sg, h in old_handlers.items():
            signal.signal(sg, h)
    if line is None:                                            # the job ended after the headline run but before its last line
        try:
            d = json.load(open(line_file))
            d["incomplete"] = "the ranks ended (rc %s%s) after the headline run and before the final line; this is the line as of the headline run" % (
                rc, ", " + state["why"] if state["why"] else "")
            line = json.dumps(d)
        except (OSError, ValueError):
            pass
    try:
        os.remove(line_file)
    except OSError:
        pass
    if line is not None:
        print(line, flush=True)
    else:
        sys.stderr.write("bench.py: the %d-rank job printed no line (rc %s%s)\n" % (args.gpus, rc, ", " + state["why"] if state["why"] else ""))
    if state["why"] and state["why"].startswith("the parent was stopped"):
        return 130
    return rc if rc else (0 if line is not None else 1)


def workload_of(n, kind=None, source=None):
    """BASELINE configs 2-5 by grid: (initial field, source-forced?, dt)."""
    kind = kind or ("gaussian" if n == 8192 else ("kuo2004" if n >= 4096 else "elliptic"))
    with_source = bool(source) if source is not None else n == 16384           # configs[4]: main-shallow-water.cpp path
    return kind, with_source, (3.0 if n <= 1024 else 3.0 * 1024 / n)


def workload_text(n, kind, with_source, dt):
    return "%dx%d %s initial field%s, nu=6.5, L=600 km, dt=%.4g s, 4 RK stages/step, fused HIP path" % (
        n, n, kind, ", source-forced (main-shallow-water.cpp path: vort_src = the FIFO producer's cake, in force every step)" if with_source else "", dt)


def timed_steps(model, reset_state, K, W, spinup, barrier, sync):
    """Spin-up (untimed, state restored), W warm-up steps, then EXACTLY K steps between barrier + synchronize pairs: seconds."""
    # Device spin-up (untimed, state restored afterwards): after >= 20 ms of idleness this GPU needs ~25 ms of
    # *this* load to return to full speed (tools/step_trend.py: 1.31 -> 1.18 ms/step over the first 20 steps; a
    # bandwidth-only torch kernel does not trigger it).  A short warm-up (W <= 10 steps) would otherwise be
    # timed on that ramp.  The spin-up runs the workload itself and then resets the state, so the W warm-up
    # steps and the K timed steps start from the initial condition as usual.
    if spinup > 0:
        model.step(spinup)
        reset_state()
    model.step(W)
    sync()
    barrier()
    sync()
    t0 = time.perf_counter()
    model.step(K)
    sync()
    barrier()
    sync()
    return time.perf_counter() - t0


def single_gpu_run(X, torch, n, kind, with_source, dt, K, W, spinup=None):
    """One GPU, the fused model: (model, seconds for K steps, spin-up steps used)."""
    model = X.Model(n, n, dt=dt)
    v0 = torch.from_numpy(X.make_field(kind, n)).cuda()      # device copy: resetting the state does not idle the GPU
    if with_source:                                          # vort_src as the FIFO producer hands it over (vort_src_input.cpp:35-46), in force for every step
        model.set_source(X.make_source_kuo2004(n))
    model.set_vort(v0)
    if spinup is None:                                       # ~30 ms of work at the single-GPU rate of the grid
        spinup = max(2, min(200, int(24 * (4096.0 / n) ** 2)))
    elapsed = timed_steps(model, lambda: model.set_vort(v0), K, W, spinup, lambda: None, torch.cuda.synchronize)
    return model, elapsed, spinup


def slab_run(X, torch, dist, slab, args, n, kind, with_source, dt, rank, world):
    """`world` ranks, the engine-driven slab model on grid n: dict of what the line reports about it (identical on every rank)."""
    K, W = args.steps, args.warmup
    dev = "cuda" if args.backend == "nccl" else "cpu"
    # the engine-driven model: local passes, exchange buffers and the RCCL all-to-all transposes behind the C ABI
    model = slab.SlabModel(n, n, dt=dt, rank=rank, world=world)
    info = {"rows_per_rank": model.XL, "active_cols_per_rank": model.KA, "frozen_cols_per_rank": model.KF,
            "field_groups": model.field_groups, "row_chunks": model.row_chunks, "col_groups": model.col_groups,
            "transport": model.transport}
    if args.backend == "nccl" and not model.transport.startswith("rccl (engine"):
        raise SystemExit("bench.py: the multi-GPU line is only printed for the engine's RCCL transport, got %r" % model.transport)
    # start-up check of the links before anything is timed: a known pattern through the transport, every word verified
    wrong = model.transport_selftest()
    tw = torch.tensor([wrong], device=dev, dtype=torch.int64)
    dist.all_reduce(tw, op=dist.ReduceOp.SUM)
    if int(tw.item()) != 0:
        raise SystemExit("bench.py: the transport self-test found %d wrong words (rank %d: %d)" % (int(tw.item()), rank, wrong))
    info["transport_selftest"] = "ok"
    # what the communicator itself reports on every rank (ncclCommCount / ncclCommUserRank / ncclCommCuDevice) and the HIP device ordinals
    mine = model.transport_info()
    mine["rank"] = rank
    every = [None] * world
    dist.all_gather_object(every, mine)
    every.sort(key=lambda d: d["rank"])
    counts = [d["comm_ranks"] for d in every]
    rccl_ranks = counts[0] if (min(counts) == max(counts) and counts[0] > 0) else None
    if args.backend == "nccl" and max(counts) > 0 and rccl_ranks != world:      # (-1 everywhere: this RCCL build lacks ncclCommCount; reported as null)
        raise SystemExit("bench.py: ncclCommCount reports %r on the ranks, expected %d everywhere" % (counts, world))
    v0_local = torch.from_numpy(slab.local_rows(X.make_field(kind, n), rank, world)).cuda()
    src_local = torch.from_numpy(slab.local_rows(X.make_source_kuo2004(n), rank, world)).cuda() if with_source else None
    if with_source:
        model.set_source_local(src_local)
    model.set_vort_local(v0_local)
    # spin-up for N > 1 as well: ~40 ms of this load at the rate the model of DESIGN.md section 6 predicts
    pred_ms0, _ = slab.predicted_step_ms(n, n, world)
    spinup = args.spinup_steps if args.spinup_steps is not None else max(2, min(100, int(40.0 / pred_ms0)))
    elapsed = timed_steps(model, lambda: model.set_vort_local(v0_local), K, W, spinup, dist.barrier, torch.cuda.synchronize)
    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    model.close()
    # how much of that is local work: the same schedule with a transport that moves nothing (fields are garbage, timing is not)
    m0 = slab.EngineSlab(n, n, dt=dt, rank=rank, world=world, transport="null", dist=dist)
    m0.set_vort_local(v0_local)
    if with_source:
        m0.set_source_local(src_local)
    m0.step(max(1, min(W, 3)))
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    m0.step(K)
    torch.cuda.synchronize()
    tl = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
    dist.all_reduce(tl, op=dist.ReduceOp.MAX)
    local_ms = 1e3 * float(tl.item()) / K
    m0.close()
    del v0_local, src_local
    pred_ms, terms = slab.predicted_step_ms(n, n, world, local_ms)
    return {"steps_per_s": K / elapsed, "ms_per_step": 1e3 * elapsed / K, "spinup_steps": spinup, "slab": info,
            "local_passes_ms_per_step": local_ms,          # the step with exchanges that move nothing: what is left is the links
            "rccl_ranks": rccl_ranks, "devices": [d["hip_device"] for d in every],
            "rccl": {"comm_ranks": counts, "comm_rank": [d["comm_rank"] for d in every], "comm_device": [d["comm_device"] for d in every]},
            "predicted": dict({"steps_per_s": 1e3 / pred_ms, "what": "DESIGN.md section 6 model (slab.predicted_step_ms): stage = link time of "
                               "5 fields x XL x KA complex per peer at an ASSUMED rate + the local work no transfer hides; not a measurement"},
                              **terms)}


def one_gpu_same_grid(X, torch, dist, args, n, kind, with_source, dt, rank):
    """Rank 0 alone runs the single-GPU model on the same grid in the same job (the other ranks wait): steps/s, same on every rank."""
    dev = "cuda" if args.backend == "nccl" else "cpu"
    rate = 0.0
    if rank == 0:
        try:
            model, elapsed, _ = single_gpu_run(X, torch, n, kind, with_source, dt, args.steps, args.warmup)
            rate = args.steps / elapsed
            del model
            torch.cuda.empty_cache()
        except Exception as e:                                   # reported, not fatal: the multi-GPU figure stands on its own
            sys.stderr.write("bench.py: single-GPU run of %d^2 on rank 0 failed: %r\n" % (n, e))
    t = torch.tensor([rate], device=dev, dtype=torch.float64)
    dist.broadcast(t, src=0)
    return float(t.item()) or None


def _driver_breakdown(lines):
    m = re.search(r"of those ([0-9.]+) s: ([0-9.]+) s stepping, ([0-9.]+) s with a record step holding the compute stream .*?, ([0-9.]+) s between the last "
                  r"step's end and the last file \(writer tail\), (-?[0-9.]+) s unaccounted", lines[2]) if len(lines) > 2 else None
    if not m:
        return None
    return dict(zip(("wall", "stepping", "record_steps_on_compute_stream", "writer_tail", "unaccounted"), (float(m.group(i)) for i in range(1, 6))))


def driver_leg(X, n, kind, with_source, dt, steps, record_step=100):
    """The drop-in C++ driver itself (host/barotropic_main.out: main.cpp:65-328 on the engine) on the same workload, BASELINE.md's stated run:
    `steps` RK4 steps with a record every `record_step` (configuration.hpp:34-36), five record files per record step written by the
    driver's writer thread into a temporary directory.  Parses the driver's own [timing] lines (stderr)."""
    import shutil
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "xlab-fftbarotropic_amd", "host", "barotropic_main.out")
    if not os.access(exe, os.X_OK):
        return {"error": "host/barotropic_main.out is not built"}
    d = tempfile.mkdtemp(prefix="bench_driver_")
    try:
        os.makedirs(os.path.join(d, "input"))
        os.makedirs(os.path.join(d, "output"))
        X.make_field(kind, n).tofile(os.path.join(d, "input", "initial_vorticity.bin"))
        cmd = [exe, "--npts", str(n), "--dt", repr(dt), "--steps", str(steps), "--record-step", str(record_step)]
        if with_source:
            return {"error": "the driver leg runs the plain driver only (no FIFO producer is started here)"}
        t0 = time.perf_counter()
        res = subprocess.run(cmd, cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, timeout=900)
        wall = time.perf_counter() - t0
        if res.returncode != 0:
            return {"error": "driver rc %d: %s" % (res.returncode, res.stderr[-300:])}
        lines = [ln for ln in res.stderr.splitlines() if ln.startswith("[timing]")]
        m1 = re.search(r"without the record steps ([0-9.]+) steps/s \(([0-9.]+) ms/step\) = ([0-9.]+) GB/s", lines[0]) if lines else None
        m2 = re.search(r"with (\d+) record steps \(([0-9.]+) GB written[^)]*\): ([0-9.]+) steps/s over ([0-9.]+) s of wall time; the writer thread was busy "
                       r"([0-9.]+) s = ([0-9.]+) GB/s", lines[1]) if len(lines) > 1 else None
        if not (m1 and m2):
            return {"error": "no [timing] lines from the driver", "stderr_tail": res.stderr[-300:]}
        return {"program": "xlab-fftbarotropic_amd/host/barotropic_main.out " + " ".join(cmd[1:]),
                "steps": steps, "record_step": record_step, "records": int(m2.group(1)),
                "driver_steps_per_s": float(m1.group(1)),              # the step loop alone: GPU time between the record steps (events on the compute stream)
                "driver_ms_per_step": float(m1.group(2)), "driver_achieved_hbm_GBs": float(m1.group(3)),
                "driver_wall_steps_per_s": float(m2.group(3)),         # host clock around the loop, writer thread and its files included
                "driver_wall_s": float(m2.group(4)), "record_GB_written": float(m2.group(2)),
                "writer_busy_s": float(m2.group(5)), "writer_disk_GBs": float(m2.group(6)),
                "process_wall_s": wall,                                # start-up (context, tables, pitch probe), input read and teardown included
                "wall_breakdown_s": _driver_breakdown(lines),          # the driver's own account of where the wall time went (third [timing] line)
                "timing_lines": lines}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # before any GPU call and before torch is imported: the ranks are a child job
        raise SystemExit(self_launch(args))

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("bench.py --gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d (or from a plain shell, "
                         "which starts the ranks itself)" % (args.gpus, world, args.gpus))
    torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))     # rehearsal on one GPU: ranks share it

    import xlab_fftbarotropic_amd as X
    n = args.n
    kind, with_source, dt = workload_of(n, args.kind, args.source)
    K, W = args.steps, args.warmup
    alg_bytes = 320.0 * n * n

    def base_line(steps_per_s, ms_per_step, spinup):
        return {
            "metric": "RK4 steps/sec, %dx%d periodic grid" % (n, n), "value": steps_per_s, "unit": "steps/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "spinup_steps": spinup,
            "config": {"workload": workload_text(n, kind, with_source, dt),
                       "grid": [n, n], "source": with_source, "parallelism": "slab%d" % world if world > 1 else "single"},
            "achieved_hbm_GBs": alg_bytes * steps_per_s / 1e9,
            "step_roofline_frac": alg_bytes * steps_per_s / 1e9 / (HBM_PEAK_GBS * world),
            "tight_frac": 256.0 * n * n * steps_per_s / 1e9 / (HBM_PEAK_GBS * world),
        }

    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
        else:
            dist.init_process_group(args.backend)
        from importlib import import_module
        slab = import_module("xlab-fftbarotropic_amd.slab")
        r = slab_run(X, torch, dist, slab, args, n, kind, with_source, dt, rank, world)
        out = base_line(r["steps_per_s"], r["ms_per_step"], r["spinup_steps"])
        out["config"]["slab"] = r["slab"]
        for k in ("local_passes_ms_per_step", "rccl_ranks", "devices", "rccl", "predicted"):
            out[k] = r[k]
        leave = os.environ.get("FB_BENCH_LINE_FILE")
        if rank == 0 and leave:                                  # a later failure must not cost the headline figure (self_launch reads this)
            with open(leave, "w") as f:
                json.dump(out, f)
        one = one_gpu_same_grid(X, torch, dist, args, n, kind, with_source, dt, rank)
        out["one_gpu_same_grid_steps_per_s"] = one               # rank 0 alone, the fused single-GPU model, same grid, same job
        out["vs_1gpu_same_grid"] = (r["steps_per_s"] / one) if one else None
        # the workloads BASELINE names for this many GPUs, in the same job: configs[3] at N = 4, configs[4] at N = 8
        also = {"auto": {4: [8192], 8: [16384]}.get(world, []), "none": []}.get(args.also_grid)
        if also is None:
            also = [int(g) for g in args.also_grid.split(",") if g]
        out["configs_run"] = []
        for g in also:
            gk, gs, gdt = workload_of(g)
            rg = slab_run(X, torch, dist, slab, args, g, gk, gs, gdt, rank, world)
            og = one_gpu_same_grid(X, torch, dist, args, g, gk, gs, gdt, rank)
            out["configs_run"].append({
                "workload": workload_text(g, gk, gs, gdt), "grid": [g, g], "source": gs, "value": rg["steps_per_s"], "unit": "steps/s",
                "ms_per_step": rg["ms_per_step"], "steps": K, "warmup": W, "spinup_steps": rg["spinup_steps"],
                "local_passes_ms_per_step": rg["local_passes_ms_per_step"], "predicted": rg["predicted"], "slab": rg["slab"],
                "rccl_ranks": rg["rccl_ranks"], "one_gpu_same_grid_steps_per_s": og, "vs_1gpu_same_grid": (rg["steps_per_s"] / og) if og else None,
                "step_roofline_frac": 320.0 * g * g * rg["steps_per_s"] / 1e9 / (HBM_PEAK_GBS * world)})
            if rank == 0 and leave:
                with open(leave, "w") as f:
                    json.dump(out, f)
        if rank == 0:
            print(json.dumps(out), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return

    model, elapsed, spinup = single_gpu_run(X, torch, n, kind, with_source, dt, K, W, args.spinup_steps)
    out = base_line(K / elapsed, 1e3 * elapsed / K, spinup)
    steps_per_s = K / elapsed

    # per-kernel HIP-event timing over a second pass of the same K steps (events on the launch stream)
    prof = model.profile_steps(K)
    torch.cuda.synchronize()
    if prof["k_col_strided_bwd4"][1] == 0 and prof["k_col_strided_fwd1"][1] == 0:     # single-pass x transform in use
        prof = {"k_row_fused": prof["k_row_fused"], "k_col_full": prof["k_col_mid"]}
    per = {k: (ms / max(cnt, 1)) for k, (ms, cnt) in prof.items()}
    tot = {k: ms for k, (ms, cnt) in prof.items()}
    dom = max(tot, key=tot.get)
    pmc_path, pmc = pmc_profile(n)
    traffic = {k: pmc_traffic(pmc, k) for k in per}
    ach = ALG_N2[dom] * n * n / (per[dom] * 1e-3) / 1e9
    out["roofline"] = {"kernel": dom, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": ach / HBM_PEAK_GBS, "traffic": traffic[dom],
                       "traffic_frac": (traffic[dom] / (per[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic[dom] else None,
                       # provenance of `traffic`: NOT measured by this run (PMC passes need their own rocprofv3 runs)
                       "traffic_source": ({"file": pmc_path, "commit": pmc.get("commit"), "command": pmc.get("command")} if pmc else None),
                       "avg_launch_ms": per[dom], "alg_bytes_per_launch": ALG_N2[dom] * n * n}
    out["kernels_ms_per_launch"] = per
    out["kernels_ms_per_step"] = {k: v / K for k, v in tot.items()}
    # credit-based and counter-based rates side by side: the contract's 8 N^2-per-pass credit counts the row pass's
    # traffic twice (it reads 4 C and writes 1 C for 40 N^2 of credit), the counter bytes are what really crosses the fabric
    out["kernels_GBs"] = {k: {"credit": ALG_N2[k] * n * n / (per[k] * 1e-3) / 1e9,
                              "traffic": (traffic[k] / (per[k] * 1e-3) / 1e9) if traffic[k] else None} for k in per}
    if all(traffic.values()):
        per_step = sum(traffic[k] * (prof[k][1] / K) for k in per)
        out["traffic_bytes_per_step"] = per_step
        out["traffic_frac"] = per_step * steps_per_s / 1e9 / HBM_PEAK_GBS
    del model
    torch.cuda.empty_cache()
    dsteps = args.driver_steps if args.driver_steps is not None else (1000 if (n == 4096 and not with_source) else 0)
    if dsteps > 0:
        out["driver"] = driver_leg(X, n, kind, with_source, dt, dsteps)
        if "driver_steps_per_s" in out["driver"]:
            out["driver"]["vs_value"] = out["driver"]["driver_steps_per_s"] / steps_per_s
    if args.cpu_steps > 0:
        out["cpu_baseline"] = cpu_baseline(n, dt, kind, args.cpu_steps, with_source)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
