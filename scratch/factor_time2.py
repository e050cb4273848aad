import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import medgp_amd
from medgp_amd import synth
Q, D, R, N = 3, 4, 2, 300
m, t, y = synth.patient(3, 0, D, N)
th = synth.theta(3, 0, 7, Q, D, R)
ctx = medgp_amd.Context(7, Q, D, R); ctx.reserve(256, N, 256)
ctx.set_patient(0, m, t, y)
ctx.nlml_grad([0], np.zeros((1, ctx.H)), False)
t0 = time.perf_counter(); ctx.set_prior(-1, *synth.hier_gamma_prior(Q, D, R, 0.01)); print("set_prior(-1) ms", 1e3*(time.perf_counter()-t0))
o = np.argsort(t, kind="stable")
for rep in range(3):
    t0 = time.perf_counter(); ctx.set_patient(0, m[o], t[o], y[o]); t1 = time.perf_counter()
    L, z, st = ctx.factor(0, th, N); t2 = time.perf_counter()
    print(f"set_patient {1e3*(t1-t0):.3f} ms, factor {1e3*(t2-t1):.3f} ms, status {st}")
