#!/usr/bin/env python3
"""Time of ONE rank's local passes of the multi-GPU model (null transport: the exchanges move nothing) against the single-GPU step:
tools/slab_local_time.py [n=4096] [world ...] -- ms per step of the single-GPU model and of rank 0 of world = 2, 4, 8 (or the given ones)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import xlab_fftbarotropic_amd as X
from importlib import import_module
S = import_module("xlab-fftbarotropic_amd.slab")

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
out = {"grid": n}
v0 = X.make_field("kuo2004", n)
m = X.Model(n, n, dt=3.0 * 1024 / n)
m.set_vort(v0); m.step(30); m.fop.synchronize()
out["single_gpu_ms"] = m.time_steps(30) / 30
m.close()
worlds = [int(a) for a in sys.argv[2:]] or [2, 4, 8]
for world in worlds:
    e = S.EngineSlab(n, rank=0, world=world, transport="null", dt=3.0 * 1024 / n)
    xl = n // world
    e.set_vort_local(v0[:xl])
    e.step(10); e.synchronize()
    t = e.time_steps(20) / 20
    out["world%d_local_ms" % world] = t
    import time
    e.synchronize()
    t0 = time.perf_counter(); e.step(20); t1 = time.perf_counter(); e.synchronize(); t2 = time.perf_counter()
    out["world%d_host_enqueue_ms" % world] = (t1 - t0) / 20 * 1e3      # host time to enqueue one step (the null transport's Python callback included)
    out["world%d_wall_ms" % world] = (t2 - t0) / 20 * 1e3
    out["world%d_plan" % world] = [e.field_groups, e.row_chunks]
    out["world%d_ideal_ms" % world] = out["single_gpu_ms"] / world
    e.close()
print(json.dumps(out))
