import sys, os, ctypes as C, numpy as np
sys.path.insert(0, '/root/repo')
import medgp_amd
from medgp_amd import capi, synth
capi.lib_path = lambda: '/root/repo/scratch/libmedgp_hip_stamps.so'
D,N,Q,R,P=24,512,5,8,512
pts, th = synth.cohort(11, 16, D, N, Q=Q, R=R)
ctx = medgp_amd.Context(7, Q, D, R); ctx.reserve(P, N, P)
for s in range(P): ctx.set_patient(s, *pts[s % 16])
th = np.stack([th[s % 16] for s in range(P)])
lib=capi.load(); lib.medgp_debug_read_xk.argtypes=[C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int]
ctx.nlml_grad(np.arange(P), th, True)
buf=np.zeros(4,np.uint64); lib.medgp_debug_read_xk(ctx._h, 3, buf.ctypes.data_as(C.c_void_p), 32, 1)
ctx.nlml_grad(np.arange(P), th, True)
lib.medgp_debug_read_xk(ctx._h, 3, buf.ctypes.data_as(C.c_void_p), 32, 0)
a=buf.astype(np.float64); tot=a[:3].sum()
print(f"k_wgrad patient 3: {int(a[3])} waves; per-wave mean cycles {tot/a[3]:.0f}; phase1(MFMA+loads) {100*a[0]/tot:.0f}%  phase2(tile->LDS) {100*a[1]/tot:.0f}%  phase3(elementwise+bins) {100*a[2]/tot:.0f}%")
