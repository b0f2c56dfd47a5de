#!/usr/bin/env python3
"""Generates the committed fixtures in tests/golden/.  Run in the build container only.

Two provenances, kept apart:
  ref_*   : outputs of the REFERENCE's own sources compiled unmodified into oracle/_ref
            (fieldio.cpp, makefield-*.cpp, vort_src_input.cpp -- the parts that need no FFTW),
            at the reference's compiled-in NPTS=768 (configuration.hpp:18).  Data only.
  fp64_*  : outputs of tests/ref_numpy.py, an independent fp64 numpy restatement of
            main.cpp:146-317.  NOT reference outputs: the reference's FFT-dependent programs
            cannot be built here (no FFTW) and the reference ships no fixtures, so these pin
            the oracle to the mathematical definition only ("parity unpinned" by the reference).
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import ref_numpy as R  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")
N_REF = 768


def sha(b):
    return hashlib.sha256(b).hexdigest()


def ref_generators():
    out = {}
    arrays = {}
    with tempfile.TemporaryDirectory() as d:
        os.mkdir(os.path.join(d, "input"))
        for exe, kind in (("makefield-elliptic-vortex", "elliptic"), ("makefield-Kuo2004", "kuo2004"),
                          ("makefield-gaussian", "gaussian"), ("makefield-const-vortex", "const")):
            subprocess.check_call([os.path.join(REF, exe + ".out")], cwd=d, stderr=subprocess.DEVNULL)
            raw = open(os.path.join(d, "input", "initial_vorticity.bin"), "rb").read()
            f = np.frombuffer(raw, dtype="<f4").reshape(N_REF, N_REF)
            out[kind] = {"sha256": sha(raw), "nbytes": len(raw), "sum64": float(f.astype(np.float64).sum()),
                         "max": float(f.max())}
            arrays["ref_gen_" + kind + "_sub16"] = f[::16, ::16].copy()
        # FIFO producer byte stream (vort_src_input.cpp:35-61) at the compiled-in configuration
        raw = subprocess.run([os.path.join(REF, "vort_src_input.out")], cwd=d, stdout=subprocess.PIPE,
                             stderr=subprocess.DEVNULL, check=True).stdout
        out["fifo_stream"] = {"sha256": sha(raw), "nbytes": len(raw), "n_flag1": int(sum(1 for b in raw if b == 1))}
    return out, arrays


def fp64_vectors():
    N = 64
    L = 600000.0
    arrays = {}
    rng = np.random.default_rng(20261004)
    spec = (rng.standard_normal((N, N // 2 + 1)) + 1j * rng.standard_normal((N, N // 2 + 1))).astype(np.complex64)
    arrays["fp64_spec_in"] = spec
    gx, gy, lap, lapi, mask = R.tables(N, N, L, L)
    s = spec.astype(np.complex128)
    arrays["fp64_gradx"] = (1j * gx.astype(np.float64)[:, None] * s)
    arrays["fp64_grady"] = (1j * gy.astype(np.float64)[None, :] * s)
    arrays["fp64_laplacian"] = s * lap.astype(np.float64)
    arrays["fp64_invlap"] = s / lapi.astype(np.float64)
    arrays["fp64_dealiase"] = s * mask.astype(np.float64)
    # model: elliptic vortex rescaled to fit a 64x64 grid is simply the generator at N=64
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    v0 = O.make_field("elliptic", N)          # generator pinned bit-exact against oracle/_ref at 768
    arrays["fp64_vort0"] = v0
    m = R.Model64(N, N, L, L, 6.5, 3.0)
    m.set_vort(v0)
    psi, u, v = m.diag()
    arrays["fp64_psi0"], arrays["fp64_u0"], arrays["fp64_v0"] = psi, u, v
    done = 0
    for upto in (1, 10, 100):
        m.step(upto - done)
        done = upto
        arrays["fp64_vort_step%d" % upto] = m.vort()
    return arrays


def main():
    meta, arrays = ref_generators()
    arrays.update(fp64_vectors())
    np.savez_compressed(os.path.join(HERE, "golden.npz"), **arrays)
    json.dump(meta, open(os.path.join(HERE, "ref_meta.json"), "w"), indent=1, sort_keys=True)
    print("wrote", os.path.join(HERE, "golden.npz"), os.path.getsize(os.path.join(HERE, "golden.npz")), "bytes")


if __name__ == "__main__":
    main()
