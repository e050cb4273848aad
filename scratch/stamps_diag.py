import sys, os, ctypes as C, numpy as np
sys.path.insert(0, '/root/repo')
os.environ['MEDGP_DBG_NOWGRAD']='1'
import medgp_amd
from medgp_amd import capi, synth
capi.lib_path = lambda: os.environ.get('STAMP_LIB', '/root/repo/scratch/libmedgp_hip_stamps.so')
D,N,Q,R=24,512,5,8
P=int(os.environ.get("SP","512"))
pts, th = synth.cohort(11, 16, D, N, Q=Q, R=R)
ctx = medgp_amd.Context(7, Q, D, R); ctx.reserve(P, N, P)
for s in range(P): ctx.set_patient(s, *pts[s % 16])
th = np.stack([th[s % 16] for s in range(P)])
lib=capi.load()
buf=np.zeros(8,np.uint64)
ctx.nlml_grad(np.arange(P), th, True)
lib.medgp_debug_read_diag(buf.ctypes.data_as(C.c_void_p))
ctx.nlml_grad(np.arange(P), th, True)
lib.medgp_debug_read_diag(buf.ctypes.data_as(C.c_void_p))
a=buf.astype(np.float64); n=a[7]
names=['diag16 x4','panel tiles','trailing tiles','inverse tiles','zero+log']
print(f"P={P}: {int(n)} block factorisations; cycles per factorisation:", {names[i]: int(a[i]/n) for i in range(5)}, 'total', int(a[:5].sum()/n))
