import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import medgp_amd
from medgp_amd import synth
def run(D,N,Q,R,P):
    pts, th = synth.cohort(11, min(P,8), D, N, Q=Q, R=R)
    ctx = medgp_amd.Context(7, Q, D, R); ctx.reserve(P, N, P)
    for s in range(P): ctx.set_patient(s, *pts[s % len(pts)])
    th = np.stack([th[s % len(pts)] for s in range(P)])
    ctx.nlml_grad(np.arange(P), th, True)
    ctx.profile_enable(True); ctx.nlml_grad(np.arange(P), th, True); prof={k:round(v[0],3) for k,v in ctx.profile_read().items() if v[1]>0}
    print(f"D{D} N{N} P{P}: {prof} total {sum(prof.values()):.2f} ms", flush=True)
    ctx.close()
for (N,P) in ((2048,16),(2048,4),(1024,64),(1024,16),(768,32),(512,64),(512,256)): run(24,N,5,8,P)
