#!/usr/bin/env python3
"""Bitwise comparison of two library builds on a noisy 4096^2 (or given) state: tools/cmp_variants.py <alt.so|-> [n=4096] [steps=3] [KEY=VAL ...]
Runs each build in its own process (FFTBARO_LIB; "-" = the in-tree library both times) and compares vort / spectrum bit for bit;
KEY=VAL pairs are set in the environment of the second run only (run-time switches)."""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %r)
import xlab_fftbarotropic_amd as X
n, steps = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(5)
v0 = (rng.standard_normal((n, n)) * 1e-4).astype(np.float32) + X.make_field("kuo2004", n)
m = X.Model(n, n, dt=3.0 * 1024 / n)
m.set_vort(v0)
m.step(steps)
np.savez(sys.argv[3], vort=m.vort().cpu().numpy(), spec=m.spectrum().cpu().numpy())
''' % ROOT
alt = None if sys.argv[1] == "-" else os.path.abspath(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096; steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
extra = dict(a.split("=", 1) for a in sys.argv[4:])
with tempfile.TemporaryDirectory() as d:
    outs = []
    for tag, lib in (("base", None), ("alt", alt)):
        e = dict(os.environ)
        e.pop("FFTBARO_LIB", None)
        if lib:
            e["FFTBARO_LIB"] = lib
        if tag == "alt":
            e.update(extra)
        o = os.path.join(d, tag + ".npz")
        subprocess.check_call([sys.executable, "-c", CHILD, str(n), str(steps), o], env=e)
        outs.append(np.load(o))
    for k in ("vort", "spec"):
        a, b = outs[0][k], outs[1][k]
        same = np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))
        print(k, "bitwise equal" if same else "DIFFERENT: max abs diff %g" % np.abs(a - b).max())
