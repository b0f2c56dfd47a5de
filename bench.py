#!/usr/bin/env python3
"""bench.py -- RK4 steps/s of the pseudospectral barotropic-vorticity hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

One "step" = one RK4 step (4 stages = 16 c2r + 4 r2c 2-D FFTs + all pointwise work of
main.cpp:286-317) of the synthetic workload below, state resident in HBM.  Prints ONE JSON line.

Workload at N=1: BASELINE.json configs[2] -- 4096x4096 Kuo2004 initial field, fp32,
dt = 3*1024/4096 s (the reference's dt=3 s is unstable above N~2300).
For N>1 the same grid is split into x-row / ky-column slabs (strong scaling), see DESIGN.md.
"""
import argparse
import json
import os
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
# HBM bytes per launch from the PMC passes committed in profiles/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
# separate runs, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950); only known for the 4096^2 run
PMC_KERNEL = {"k_col_strided_bwd4": "k_col_strided<64, 1>", "k_row_fused": "k_row8<false>",
              "k_col_strided_fwd1": "k_col_strided<64, -1>", "k_col_mid": "k_col_mid<64>", "k_col_full": "k_col_full"}


def pmc_traffic(kernel, n):
    path = os.path.join(ROOT, "profiles", "r01_i_pmc_traffic_4096.json")
    if n != 4096 or not os.path.exists(path):
        return None
    try:
        ks = json.load(open(path))["kernels"]
        if kernel == "k_col_full":                  # four template instances (one per RK stage): average them
            v = [d["hbm_bytes_per_launch_corrected"] for k, d in ks.items() if k.startswith("k_col_full")]
            return sum(v) / len(v) if v else None
        return ks[PMC_KERNEL[kernel]]["hbm_bytes_per_launch_corrected"]
    except (KeyError, ValueError):
        return None


def cpu_baseline(n, dt, kind, steps):
    """Oracle ("port") timed on ONE host core -- the reference is single-threaded."""
    import ctypes
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    try:
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(1)
    except OSError:
        pass
    m = O.Model(n, n, dt=dt)
    m.set_vort(O.make_field(kind, n))
    m.step(1)
    t0 = time.perf_counter()
    m.step(steps)
    el = time.perf_counter() - t0
    return {"value": steps / el, "unit": "steps/s", "cores": 1, "kind": "port",
            "sample": "%dx%d %s, %d RK4 steps after 1 warm-up, oracle/liboracle.so (own FFT, reference loop structure), 1 thread"
                      % (n, n, kind, steps)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--grid", "--n", dest="n", type=int, default=4096, help="grid points per side")
    ap.add_argument("--kind", default=None, help="initial field (default: kuo2004 for n>=4096 else elliptic)")
    ap.add_argument("--cpu-steps", type=int, default=2, help="oracle steps for cpu_baseline (0 = skip)")
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
    kind = args.kind or ("kuo2004" if n >= 4096 else "elliptic")
    dt = 3.0 if n <= 1024 else 3.0 * 1024 / n
    K, W = args.steps, args.warmup

    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
        else:
            dist.init_process_group(args.backend)
        from importlib import import_module
        slab = import_module("xlab-fftbarotropic_amd.slab")
        model = slab.SlabModel(n, n, dt=dt, rank=rank, world=world)
        v0_local = slab.local_rows(X.make_field(kind, n), rank, world)

        def reset_state():
            model.set_vort_local(v0_local)
        reset_state()
        barrier = dist.barrier
    else:
        model = X.Model(n, n, dt=dt)
        v0 = torch.from_numpy(X.make_field(kind, n)).cuda()      # device copy: resetting the state does not idle the GPU

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
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ms_per_step = 1e3 * elapsed / K
    steps_per_s = K / elapsed
    alg_bytes = 320.0 * n * n
    out = {
        "metric": "RK4 steps/sec, %dx%d periodic grid" % (n, n), "value": steps_per_s, "unit": "steps/s",
        "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "spinup_steps": args.spinup_steps,
        "config": {"workload": "%dx%d %s initial field, nu=6.5, L=600 km, dt=%.4g s, 4 RK stages/step, fused HIP path"
                               % (n, n, kind, dt), "grid": [n, n], "parallelism": "slab%d" % world if world > 1 else "single"},
        "achieved_hbm_GBs": alg_bytes * steps_per_s / 1e9,
        "step_roofline_frac": alg_bytes * steps_per_s / 1e9 / (HBM_PEAK_GBS * world),
    }

    if rank == 0 and world == 1:
        # per-kernel HIP-event timing over a second pass of the same K steps (events on the launch stream)
        prof = model.profile_steps(K)
        torch.cuda.synchronize()
        if prof["k_col_strided_bwd4"][1] == 0 and prof["k_col_strided_fwd1"][1] == 0:     # single-pass x transform in use
            prof = {"k_row_fused": prof["k_row_fused"], "k_col_full": prof["k_col_mid"]}
        per = {k: (ms / max(cnt, 1)) for k, (ms, cnt) in prof.items()}
        tot = {k: ms for k, (ms, cnt) in prof.items()}
        dom = max(tot, key=tot.get)
        ach = ALG_N2[dom] * n * n / (per[dom] * 1e-3) / 1e9
        out["roofline"] = {"kernel": dom, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": ach / HBM_PEAK_GBS, "traffic": pmc_traffic(dom, n),
                           "avg_launch_ms": per[dom], "alg_bytes_per_launch": ALG_N2[dom] * n * n}
        out["kernels_ms_per_launch"] = per
        out["kernels_ms_per_step"] = {k: v / K for k, v in tot.items()}
        if args.cpu_steps > 0:
            out["cpu_baseline"] = cpu_baseline(n, dt, kind, args.cpu_steps)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
