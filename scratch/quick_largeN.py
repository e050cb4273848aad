import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import medgp_amd
from medgp_amd import synth
from oracle import oracle as O
def run(D,N,Q,R,P,check=False):
    pts, th = synth.cohort(11, P, D, N, Q=Q, R=R)
    ctx = medgp_amd.Context(7, Q, D, R); ctx.reserve(P, N, P)
    for s,(m,t,y) in enumerate(pts): ctx.set_patient(s, m, t, y)
    nlml,grad,st=ctx.nlml_grad(np.arange(P), th, True)
    msg=''
    if check:
        m,t,y=pts[0]; t0=time.time(); ref=O.nlml_grad(7,Q,D,R,m,t,y,th[0],nthreads=16); dt=time.time()-t0
        gs=np.abs(ref['grad']).max()
        msg=f"nlml {abs(nlml[0]-ref['nlml'])/abs(ref['nlml']):.1e} grad {(np.abs(grad[0]-ref['grad'])/np.maximum(np.abs(ref['grad']),1e-3*gs)).max():.1e} (oracle {dt:.1f}s)"
    ctx.profile_enable(True); ctx.nlml_grad(np.arange(P), th, True); prof={k:round(v[0],3) for k,v in ctx.profile_read().items() if v[1]>0}
    print(f"D{D} N{N} P{P}: st {st[:2]} {msg} | {prof} total {sum(prof.values()):.2f} ms", flush=True)
    ctx.close()
run(2,256,5,2,256)      # config 2
run(24,2048,5,8,1,True) # config 3
run(64,1024,5,8,2,True) # config 5 reduced N
run(64,4096,5,8,1)      # config 5
run(24,2048,5,8,16)
