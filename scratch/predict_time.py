import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import medgp_amd
from medgp_amd import synth
Q, D, R, N = 3, 4, 2, 300
m, t, y = synth.patient(3, 0, D, N)
th = synth.theta(3, 0, 7, Q, D, R)
P = 256
ctx = medgp_amd.Context(7, Q, D, R); ctx.reserve(P, N, P)
o = np.argsort(t, kind="stable")
ms, ts, ys = m[o], t[o], y[o]
pts = [(ms[:k], ts[:k], ys[:k]) for k in range(N - P, N)]
thb = np.repeat(th[None, :], P, 0)
m2 = ms[N - P:N].copy(); t2 = ts[N - P:N].copy()
for rep in range(4):
    t0 = time.perf_counter(); ctx.set_patients(np.arange(P), pts); t1 = time.perf_counter()
    mean, var, st = ctx.fit_predict_batch(np.arange(P), thb, m2, t2); t2_ = time.perf_counter()
    print(f"set_patients {1e3*(t1-t0):.3f} ms (incl. numpy packing), fit_predict_batch {1e3*(t2_-t1):.3f} ms, st ok {np.all(st==0)}")
ctx.profile_enable(True); ctx.fit_predict_batch(np.arange(P), thb, m2, t2); print({k: round(v[0],3) for k,v in ctx.profile_read().items() if v[1]>0})
