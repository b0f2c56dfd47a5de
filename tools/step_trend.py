"""Per-step wall time of the first steps after set_vort, and after pauses (design probe): where does the slow start come from?"""
import sys, time, numpy as np
sys.path.insert(0, sys.argv[1])
import torch, xlab_fftbarotropic_amd as X
n = 4096
m = X.Model(n, n, dt=0.75); m.set_vort(X.make_field("kuo2004", n))
def batch(label, pairs=16):
    ts = []
    for i in range(pairs):
        t0 = time.perf_counter(); m.step(2); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 500)
    print("%-26s ms/step over successive pairs of steps: %s" % (label, " ".join("%.3f" % t for t in ts)))
batch("fresh model")
batch("immediately again")
time.sleep(0.5); batch("after 0.5 s idle")
time.sleep(0.02); batch("after 20 ms idle")
time.sleep(0.002); batch("after 2 ms idle")
m2 = X.Model(n, n, dt=0.75); m2.set_vort(X.make_field("kuo2004", n))
batch("old model after creating a second one")
