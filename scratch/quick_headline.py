import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import medgp_amd
from medgp_amd import synth
from oracle import oracle as O
def run(D,N,Q,R,P,fg=True,check=1):
    pts, th = synth.cohort(11, P, D, N, Q=Q, R=R)
    ctx = medgp_amd.Context(7, Q, D, R)
    ctx.reserve(P, N, P)
    for s,(m,t,y) in enumerate(pts): ctx.set_patient(s, m, t, y)
    nlml,grad,st=ctx.nlml_grad(np.arange(P), th, fg)
    wn=wg=0
    for p in range(check):
        m,t,y=pts[p]; ref=O.nlml_grad(7,Q,D,R,m,t,y,th[p],flag_grad=fg,nthreads=8)
        wn=max(wn,abs(nlml[p]-ref['nlml'])/abs(ref['nlml']))
        if fg:
            gs=np.abs(ref['grad']).max(); wg=max(wg,(np.abs(grad[p]-ref['grad'])/np.maximum(np.abs(ref['grad']),1e-3*gs)).max())
    ctx.profile_enable(True); ctx.nlml_grad(np.arange(P), th, fg); prof={k:round(v[0],3) for k,v in ctx.profile_read().items() if v[1]>0}
    print(f"D{D} N{N} P{P} grad={fg}: nlml {wn:.1e} grad {wg:.1e} | {prof} total {sum(prof.values()):.2f} ms", flush=True)
    ctx.close()
run(24,512,5,8,8); run(24,512,5,8,256); run(24,512,5,8,512); run(24,512,5,8,512,False); run(24,256,5,8,512); run(2,256,5,2,256); run(24,700,5,8,4)
