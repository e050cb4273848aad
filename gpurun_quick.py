import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import medgp_amd
from medgp_amd import synth
from oracle import oracle as O
def check(D,N,Q,R,P,interleave=False,kidx=7,prior=False):
    pts, th = synth.cohort(11, P, D, N, kernel_index=kidx, Q=Q, R=R, interleave=interleave)
    ctx = medgp_amd.Context(kidx, Q, D, R)
    ctx.reserve(P, N, P)
    for s,(m,t,y) in enumerate(pts): ctx.set_patient(s, m if kidx==7 else None, t, y)
    pr=None
    if prior:
        f,ty,ex,p0,p1 = synth.hier_gamma_prior(Q,D,R)
        ctx.set_prior(-1,f,ty,ex,p0,p1)
        pr=O.Prior.hier_gamma(Q,D,R)
    t0=time.time(); nlml,grad,st = ctx.nlml_grad(np.arange(P), th, True); dt=time.time()-t0
    worst_n=0; worst_g=0
    for p,(m,t,y) in enumerate(pts[:4]):
        ref = O.nlml_grad(kidx,Q,D,R,m if kidx==7 else None,t,y,th[p],prior=pr,nthreads=8)
        worst_n=max(worst_n, abs(nlml[p]-ref['nlml'])/abs(ref['nlml']))
        gs=np.abs(ref['grad']).max()
        worst_g=max(worst_g, (np.abs(grad[p]-ref['grad'])/np.maximum(np.abs(ref['grad']),1e-3*gs)).max())
    print(f"kidx{kidx} D{D} N{N} P{P} il{interleave} prior{prior}: dt {dt*1e3:.1f} ms  nlml rel {worst_n:.2e} grad rel {worst_g:.2e} status {st[:4]}")
    t0=time.time(); nlml,grad,st = ctx.nlml_grad(np.arange(P), th, True); dt=time.time()-t0
    print(f"   second call {dt*1e3:.1f} ms -> {P/dt:.0f} evals/s (host-pointer API)")
    ctx.profile_enable(True); ctx.nlml_grad(np.arange(P), th, True); print('   ', {k:round(v[0],3) for k,v in ctx.profile_read().items()}); 
    ctx.close()
check(2,96,5,2,4)
check(2,150,5,2,2,interleave=True)
check(2,256,5,2,64,prior=True)
check(24,512,5,8,8)
check(24,512,5,8,64,prior=True)
check(1,200,3,0,4,kidx=8)
check(1,200,1,0,4,kidx=0)
