import sys, os, numpy as np
sys.path.insert(0, '/root/repo')
import medgp_amd
from medgp_amd import synth
def run(D,N,Q,R,P):
    pts, th = synth.cohort(11, min(P,32), D, N, Q=Q, R=R)
    ctx = medgp_amd.Context(7, Q, D, R); ctx.reserve(P, N, P)
    for s in range(P): ctx.set_patient(s, *pts[s % len(pts)])
    thx = np.stack([th[s % len(pts)] for s in range(P)])
    ctx.nlml_grad(np.arange(P), thx, True)
    ctx.profile_enable(True)
    for _ in range(3): ctx.nlml_grad(np.arange(P), thx, True)
    prof={k:round(v[0]/3,3) for k,v in ctx.profile_read().items() if v[1]>0}
    fact=sum(v for k,v in prof.items() if k in ('k_cholinv','k_ci_panel','k_ci_trsm'))
    print(f"{os.environ.get('TAG','')} N{N} P{P}: factorisation {fact:.3f} ms  total {sum(prof.values()):.2f} ms", flush=True)
    ctx.close()
for N,P in [(512,64),(512,128),(512,192),(512,256),(512,320),(256,128),(256,256),(1024,64),(1024,128),(1024,200),(2048,16),(2048,64)]:
    run(24,N,5,8,P)
