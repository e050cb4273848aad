import sys, os, numpy as np
sys.path.insert(0, '/root/repo')
import medgp_amd
from medgp_amd import capi, synth
if os.environ.get('LIB'): capi.lib_path = lambda: os.environ['LIB']
D,N,Q,R,P=24,512,5,8,512
pts, th = synth.cohort(11, 16, D, N, Q=Q, R=R)
ctx = medgp_amd.Context(7, Q, D, R); ctx.reserve(P, N, P)
for s in range(P): ctx.set_patient(s, *pts[s % 16])
th = np.stack([th[s % 16] for s in range(P)])
nl,g,st=ctx.nlml_grad(np.arange(P), th, True)
ctx.profile_enable(True)
for _ in range(5): ctx.nlml_grad(np.arange(P), th, True)
prof={k:round(v[0]/v[1],3) for k,v in ctx.profile_read().items() if v[1]>0}
print(os.environ.get('LIB','default'), 'nlml0', repr(nl[0]), 'status', st[:4], prof, flush=True)
