import sys, os, numpy as np
sys.path.insert(0, '/root/repo')
import medgp_amd
from medgp_amd import synth
from oracle import oracle as O
def run(D,N,Q,R,P,fg=True,check=2):
    pts, th = synth.cohort(11, P, D, N, Q=Q, R=R)
    ctx = medgp_amd.Context(7, Q, D, R)
    ctx.reserve(P, N, P)
    for s,(m,t,y) in enumerate(pts): ctx.set_patient(s, m, t, y)
    nlml,grad,st=ctx.nlml_grad(np.arange(P), th, fg)
    wn=wg=0
    for p in list(range(check))+[P-1]:
        m,t,y=pts[p]; ref=O.nlml_grad(7,Q,D,R,m,t,y,th[p],flag_grad=fg,nthreads=8)
        wn=max(wn,abs(nlml[p]-ref['nlml'])/abs(ref['nlml']))
        if fg:
            gs=np.abs(ref['grad']).max(); wg=max(wg,(np.abs(grad[p]-ref['grad'])/np.maximum(np.abs(ref['grad']),1e-3*gs)).max())
    ctx.profile_enable(True)
    for _ in range(3): ctx.nlml_grad(np.arange(P), th, fg)
    prof={k:round(v[0]/v[1],3) for k,v in ctx.profile_read().items() if v[1]>0}
    print(f"IL={os.environ.get('MEDGP_CHOLINV_IL')} D{D} N{N} P{P} grad={fg}: nlml {wn:.1e} grad {wg:.1e} | cholinv {prof.get('k_cholinv')} total {sum(prof.values()):.2f} ms", flush=True)
    ctx.close()
run(24,512,5,8,512); run(24,512,5,8,512,False); run(24,512,5,8,200); run(24,700,5,8,300); run(24,1024,5,8,260,check=1); run(24,256,5,8,512); run(2,256,5,2,256); run(24,100,5,8,300)
