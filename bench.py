#!/usr/bin/env python3
"""bench.py -- RK4 steps/s of the pseudospectral barotropic-vorticity hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

One "step" = one RK4 step (4 stages = 16 c2r + 4 r2c 2-D FFTs + all pointwise work of
main.cpp:286-317) of the synthetic workload below, state resident in HBM.  Prints ONE JSON line.

Workload at N=1: BASELINE.json configs[2] -- 4096x4096 Kuo2004 initial field, fp32,
dt = 3*1024/4096 s (the reference's dt=3 s is unstable above N~2300).
For N>1 the same grid is split into x-row / ky-column slabs (strong scaling; the engine drives the
RCCL all-to-all transposes itself, csrc/fb_slab_driver.h), see DESIGN.md section 6.

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


def main():
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
    args = ap.parse_args()

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                             % (args.gpus, args.gpus))
    torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))     # rehearsal on one GPU: ranks share it

    import xlab_fftbarotropic_amd as X
    n = args.n
    kind = args.kind or ("gaussian" if n == 8192 else ("kuo2004" if n >= 4096 else "elliptic"))      # BASELINE configs 2-5
    with_source = bool(args.source) if args.source is not None else n == 16384                         # configs[4]: main-shallow-water.cpp path
    dt = 3.0 if n <= 1024 else 3.0 * 1024 / n
    K, W = args.steps, args.warmup
    slab_info = None

    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
        else:
            dist.init_process_group(args.backend)
        from importlib import import_module
        slab = import_module("xlab-fftbarotropic_amd.slab")
        # the engine-driven model: local passes, exchange buffers and the RCCL all-to-all transposes behind the C ABI
        model = slab.SlabModel(n, n, dt=dt, rank=rank, world=world)
        slab_info = {"rows_per_rank": model.XL, "active_cols_per_rank": model.KA, "frozen_cols_per_rank": model.KF,
                     "field_groups": model.field_groups, "row_chunks": model.row_chunks, "col_groups": model.col_groups,
                     "transport": model.transport}
        if args.backend == "nccl" and not model.transport.startswith("rccl (engine"):
            raise SystemExit("bench.py: the multi-GPU line is only printed for the engine's RCCL transport, got %r" % model.transport)
        # start-up check of the links before anything is timed: a known pattern through the transport, every word verified
        wrong = model.transport_selftest()
        tw = torch.tensor([wrong], device="cuda" if args.backend == "nccl" else "cpu", dtype=torch.int64)
        dist.all_reduce(tw, op=dist.ReduceOp.SUM)
        if int(tw.item()) != 0:
            raise SystemExit("bench.py: the transport self-test found %d wrong words (rank %d: %d)" % (int(tw.item()), rank, wrong))
        slab_info["transport_selftest"] = "ok"
        v0_local = torch.from_numpy(slab.local_rows(X.make_field(kind, n), rank, world)).cuda()
        if with_source:
            model.set_source_local(torch.from_numpy(slab.local_rows(X.make_source_kuo2004(n), rank, world)).cuda())

        def reset_state():
            model.set_vort_local(v0_local)
        reset_state()
        barrier = dist.barrier
    else:
        model = X.Model(n, n, dt=dt)
        v0 = torch.from_numpy(X.make_field(kind, n)).cuda()      # device copy: resetting the state does not idle the GPU
        if with_source:                                          # vort_src as the FIFO producer hands it over (vort_src_input.cpp:35-46), in force for every step
            model.set_source(X.make_source_kuo2004(n))

        def reset_state():
            model.set_vort(v0)
        reset_state()

        def barrier():
            return None
    if args.spinup_steps is None:                   # ~30 ms of work at the single-GPU rate of the grid
        args.spinup_steps = 0 if world > 1 else max(2, min(200, int(24 * (4096.0 / n) ** 2)))

    # Device spin-up (untimed, state restored afterwards): after >= 20 ms of idleness this GPU needs ~25 ms of
    # *this* load to return to full speed (tools/step_trend.py: 1.31 -> 1.18 ms/step over the first 20 steps; a
    # bandwidth-only torch kernel does not trigger it).  A short warm-up (W <= 10 steps) would otherwise be
    # timed on that ramp.  The spin-up runs the workload itself and then resets the state, so the W warm-up
    # steps and the K timed steps start from the initial condition as usual.
    if args.spinup_steps > 0:
        model.step(args.spinup_steps)
        reset_state()
    model.step(W)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model.step(K)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], device="cuda" if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    local_ms = None
    if world > 1:
        # how much of that is local work: the same schedule with a transport that moves nothing (fields are garbage, timing is not)
        m0 = slab.EngineSlab(n, n, dt=dt, rank=rank, world=world, transport="null", dist=dist)
        m0.set_vort_local(v0_local)
        if with_source:
            m0.set_source_local(torch.from_numpy(slab.local_rows(X.make_source_kuo2004(n), rank, world)).cuda())
        m0.step(max(1, min(W, 3)))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        m0.step(K)
        torch.cuda.synchronize()
        tl = torch.tensor([time.perf_counter() - t1], device="cuda" if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tl, op=dist.ReduceOp.MAX)
        local_ms = 1e3 * float(tl.item()) / K
        m0.close()

    ms_per_step = 1e3 * elapsed / K
    steps_per_s = K / elapsed
    alg_bytes = 320.0 * n * n
    out = {
        "metric": "RK4 steps/sec, %dx%d periodic grid" % (n, n), "value": steps_per_s, "unit": "steps/s",
        "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "spinup_steps": args.spinup_steps,
        "config": {"workload": "%dx%d %s initial field%s, nu=6.5, L=600 km, dt=%.4g s, 4 RK stages/step, fused HIP path"
                               % (n, n, kind, ", source-forced (main-shallow-water.cpp path: vort_src = the FIFO producer's cake, in force every step)"
                                  if with_source else "", dt),
                   "grid": [n, n], "source": with_source, "parallelism": "slab%d" % world if world > 1 else "single"},
        "achieved_hbm_GBs": alg_bytes * steps_per_s / 1e9,
        "step_roofline_frac": alg_bytes * steps_per_s / 1e9 / (HBM_PEAK_GBS * world),
        "tight_frac": 256.0 * n * n * steps_per_s / 1e9 / (HBM_PEAK_GBS * world),
    }
    if slab_info:
        out["config"]["slab"] = slab_info
        out["local_passes_ms_per_step"] = local_ms          # the step with exchanges that move nothing: what is left is the links

    if rank == 0 and world == 1:
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
        if args.cpu_steps > 0:
            out["cpu_baseline"] = cpu_baseline(n, dt, kind, args.cpu_steps, with_source)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        model.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
