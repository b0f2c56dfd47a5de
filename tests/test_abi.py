"""CPU checks of the drop-in boundary: libfftbaro.so loads and exports every entry point that
include/fftbaro.h declares; no compute call is made (no GPU here).  Also guards the rule that
the product never routes through the oracle or any CPU fallback."""
import ctypes
import os
import re

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _declared():
    src = open(os.path.join(ROOT, "include", "fftbaro.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fb_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported():
    import xlab_fftbarotropic_amd as X
    L = X.lib()
    names = _declared()
    assert len(names) >= 45
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    assert set(X.EXPORTS) == set(names), set(X.EXPORTS) ^ set(names)


def test_no_gpu_means_loud_failure():
    """Without a HIP device the engine refuses to create a context (no CPU fallback)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import xlab_fftbarotropic_amd as X
    L = X.lib()
    h = ctypes.c_void_p()
    assert L.fb_create(ctypes.byref(h), 256, 256, 6e5, 6e5) == 3          # FB_EHIP
    assert b"no CPU fallback" in L.fb_last_error()
    with pytest.raises(X.FftBaroError):
        X.Model(256)
    assert L.fb_size_supported(4096, 4096) == 1 and L.fb_size_supported(768, 768) == 1
    assert L.fb_size_supported(1000, 1000) == 0 and L.fb_size_supported(768, 4096) == 1 and L.fb_size_supported(6144, 64) == 0
    assert L.fb_strerror(5) == b"unsupported grid size"


def test_product_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "xlab-fftbarotropic_amd")
    bad = []
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"oracle_py|liboracle|fb_oracle|import oracle|/root/reference", txt):
                    bad.append(os.path.join(dp, f))
    # slab.py only *mentions* the oracle in a docstring about tests
    bad = [b for b in bad if not b.endswith("slab.py")]
    assert not bad, bad


def test_host_generators_match_oracle_bitwise():
    import numpy as np
    import oracle_py as O
    import xlab_fftbarotropic_amd as X
    for kind in ("elliptic", "kuo2004", "gaussian", "const"):
        a, b = X.make_field(kind, 192), O.make_field(kind, 192)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), kind
    with pytest.raises(X.FftBaroError):
        X.make_field("nope", 64)
