import sys, os, numpy as np
sys.path.insert(0, '/root/repo')
import medgp_amd
from medgp_amd import synth
D,N,Q,R,P=24,512,5,8,512
pts, th = synth.cohort(11, P, D, N, Q=Q, R=R)
ctx = medgp_amd.Context(7, Q, D, R); ctx.reserve(P, N, P)
for s,(m,t,y) in enumerate(pts): ctx.set_patient(s, m, t, y)
nl,g,st=ctx.nlml_grad(np.arange(P), th, True)
ctx.profile_enable(True)
for _ in range(5): ctx.nlml_grad(np.arange(P), th, True)
prof={k:round(v[0]/v[1],3) for k,v in ctx.profile_read().items() if v[1]>0}
print(os.environ.get('TAG',''), 'nlml0', repr(nl[0]), prof, 'total', round(sum(prof.values()),3), flush=True)
