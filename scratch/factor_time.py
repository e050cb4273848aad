import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import medgp_amd
from medgp_amd import synth
Q, D, R, N = 3, 4, 2, int(sys.argv[1]) if len(sys.argv) > 1 else 300
m, t, y = synth.patient(3, 0, D, N, interleave=True)
th = synth.theta(3, 0, 7, Q, D, R)
ctx = medgp_amd.Context(7, Q, D, R); ctx.reserve(256, N, 256)
for rep in range(3):
    t0 = time.perf_counter(); ctx.set_patient(0, m, t, y); t1 = time.perf_counter()
    L, z, st = ctx.factor(0, th, N); t2 = time.perf_counter()
    nl, _, s2 = ctx.nlml_grad([0], th[None, :], False); t3 = time.perf_counter()
    print(f"set_patient {1e3*(t1-t0):.3f} ms, factor {1e3*(t2-t1):.3f} ms, nlml {1e3*(t3-t2):.3f} ms, status {st}")
from oracle import oracle as O
K = O.gram(7, Q, D, R, m, t, th)
print("|L L^T - K| max", np.abs(L @ L.T - K).max(), " z check", np.abs(np.linalg.solve(L, y.astype(np.float64)) - z).max())
