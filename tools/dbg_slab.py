import os, sys, numpy as np
sys.path[:0] = ["/root/repo", "/root/repo/tests"]
import xlab_fftbarotropic_amd as X
from test_gpu_slab import slab_run
for world, n, steps, usesrc in [(2, 256, 1, False), (2, 256, 1, True), (2, 256, 3, True), (4, 256, 2, True)]:
    v0 = X.make_field("elliptic", n); src = X.make_source_kuo2004(n) if usesrc else None
    ref = X.Model(n, n, dt=3.0); ref.set_vort(v0)
    if usesrc: ref.set_source(src)
    ref.step(steps); want = ref.vort().cpu().numpy()
    back, got, plan = slab_run(n, world, steps, v0, 3.0, src=src)
    d = np.fft.rfft2((got.astype(np.float64) - want))
    a = np.abs(d)
    print(world, n, steps, usesrc, plan, "max abs diff", np.abs(got - want).max(), "spectral max", a.max(), "cols with diff", np.nonzero(a.max(axis=0) > 1e-9 * a.max())[0][:20], "rows", np.nonzero(a.max(axis=1) > 1e-3 * a.max())[0][:20])
