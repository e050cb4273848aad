import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import medgp_amd
from medgp_amd import synth
D,N,Q,R,P=24,512,5,8,512
pts, th = synth.cohort(11, P, D, N, Q=Q, R=R)
ctx = medgp_amd.Context(7, Q, D, R); ctx.reserve(P, N, P)
for s,(m,t,y) in enumerate(pts): ctx.set_patient(s, m, t, y)
sl=np.arange(P)
for _ in range(3): ctx.nlml_grad(sl, th, True)
t0=time.perf_counter()
for _ in range(20): ctx.nlml_grad(sl, th, True)
dt=(time.perf_counter()-t0)/20
print(f"host-pointer API: {dt*1e3:.2f} ms per 512-patient step = {P/dt:.0f} evals/s")
