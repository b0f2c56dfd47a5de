import sys, time, torch
sys.path.insert(0, sys.argv[1])
import xlab_fftbarotropic_amd as X
for n in (256, 1024, 4096):
    for graph in (False, True):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            m = X.Model(n, n, dt=3.0 if n <= 1024 else 0.75); m.fop.use_current_stream(); m.use_graph(graph)
            m.set_vort(X.make_field("elliptic", n)); m.step(20); torch.cuda.synchronize()
            K = 400 if n < 4096 else 50
            t0 = time.perf_counter(); m.step(K); torch.cuda.synchronize(); el = time.perf_counter() - t0
        print("n=%d graph=%s: %.1f steps/s" % (n, graph, K / el))
