import sys, os, ctypes as C, numpy as np
sys.path.insert(0, '/root/repo')
os.environ['MEDGP_DBG_NOWGRAD']='1'
import medgp_amd
from medgp_amd import capi, synth
capi.lib_path = lambda: '/root/repo/scratch/libmedgp_hip_stamps.so'
D,N,Q,R=24,512,5,8
names=['init+wait','inithalf0','zsolve|diagwait','diagfac+st','mfma','trsm+end','stagest','chunkbar']
for P in (int(os.environ.get("SP","512")),):
    pts = [synth.patient(2024, s, D, N) for s in range(P)]
    th = np.stack([synth.theta(2024, s, 7, Q, D, R) for s in range(P)])
    ctx = medgp_amd.Context(7, Q, D, R); ctx.reserve(P, N, P); ctx.set_patients(np.arange(P), pts)
    slots=np.arange(P,dtype=np.int32); nl=np.empty(P); g=np.empty((P,ctx.H)); st=np.empty(P,np.int32)
    lib=capi.load()
    for it in range(2):
        st[:] = 99
        lib.medgp_nlml_grad(ctx._h, P, slots.ctypes.data_as(C.POINTER(C.c_int32)), th.ctypes.data_as(C.POINTER(C.c_double)), 1, nl.ctypes.data_as(C.POINTER(C.c_double)), g.ctypes.data_as(C.POINTER(C.c_double)), st.ctypes.data_as(C.POINTER(C.c_int32)))
    nw = 4
    print('status', np.unique(st, return_counts=True))
    for b in (0, 1, min(P-1, 40)):
        buf=np.zeros(nw*8, np.uint64)
        lib.medgp_debug_read_slab.argtypes=[C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        lib.medgp_debug_read_slab(ctx._h, b, buf.ctypes.data_as(C.c_void_p), buf.nbytes)
        a=buf.reshape(nw,8).astype(np.float64)
        tot=a.sum(axis=1)
        print(f"P={P} b={b}: total cycles/wave {tot.mean():.0f} ({tot.mean()/2.4e3:.0f} us at 2.4GHz... memtime ticks)")
        for w in range(nw):
            print('   wave',w,' '.join(f"{names[e]}={100*a[w,e]/tot[w]:.0f}%/{a[w,e]/1e3:.0f}k" for e in range(8)))
    ctx.close()
# wall time of the stamped build's kernel (HIP events) for comparison with the production build
import time
P=int(os.environ.get('SP','512'))
pts = [synth.patient(2024, s, D, N) for s in range(P)]
th = np.stack([synth.theta(2024, s, 7, Q, D, R) for s in range(P)])
ctx = medgp_amd.Context(7, Q, D, R); ctx.reserve(P, N, P); ctx.set_patients(np.arange(P), pts)
os.environ.pop('MEDGP_DBG_NOWGRAD', None)
ctx.nlml_grad(np.arange(P), th, True)
ctx.profile_enable(True)
for _ in range(3): ctx.nlml_grad(np.arange(P), th, True)
print('stamped build kernel times', {k: round(v[0]/3,3) for k,v in ctx.profile_read().items() if v[1]>0})
