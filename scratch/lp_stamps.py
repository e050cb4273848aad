"""per-step wall-clock stamps of the persistent look-ahead schedule (lib built with -DLP_STAMPS -DMEDGP_STAMPS -> scratch/lib_lpstamps.so)
usage: python scratch/lp_stamps.py N D"""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['MEDGP_LIB'] = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'lib_lpstamps.so')
os.environ['MEDGP_DBG_NOWGRAD'] = '1'
import medgp_amd
from medgp_amd import capi, synth
N, D = int(sys.argv[1]), int(sys.argv[2])
Q, R = 5, 8
m, t, y = synth.patient(11, 0, D, N); th = synth.theta(11, 0, 7, Q, D, R)
ctx = medgp_amd.Context(7, Q, D, R); ctx.reserve(1, N, 1); ctx.set_patient(0, m, t, y)
lib = capi.load()
lib.medgp_debug_read_slab.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
lib.medgp_debug_clear_slab.argtypes = [C.c_void_p, C.c_int, C.c_int]
nb = N // 64
for it in range(3):
    lib.medgp_debug_clear_slab(ctx._h, 0, 16 * (nb + 2) * 8)
    nl, g, st = ctx.nlml_grad(np.zeros(1, np.int32), th[None], True)
    buf = np.zeros(16 * (nb + 2), np.uint64)
    lib.medgp_debug_read_slab(ctx._h, 0, buf.ctypes.data_as(C.c_void_p), buf.nbytes)
a = buf.reshape(nb + 2, 16).astype(np.int64)
t0 = a[0, 0]
us = lambda x: (x - t0) / 100.0
print("chain: step | start  waited  body-end  published || key F task of the step: start  polled  acquired  body-end  drained  (us since chain start; then durations)")
for k in range(nb - 1):
    c = a[k, :4]; f = a[k, 4:9]
    cs = f"{us(c[0]):7.1f} wait {(c[1]-c[0])/100:5.1f} body {(c[2]-c[1])/100:5.1f} drain {(c[3]-c[2])/100:4.1f}"
    fs = f"{us(f[0]):7.1f} poll {(f[1]-f[0])/100:5.1f} acq {(f[2]-f[1])/100:4.1f} body {(f[3]-f[2])/100:5.1f} drain {(f[4]-f[3])/100:4.1f} -> end {us(f[4]):7.1f}" if f[0] else ""
    ls = f" || L: n {a[k,9]:4d} wait {a[k,10]/100/max(a[k,9],1):6.1f} body {a[k,11]/100/max(a[k,9],1):5.1f} drain {a[k,12]/100/max(a[k,9],1):4.1f} last end {us(a[k,13]) if a[k,13] else 0:7.1f} | F last end {us(a[k,14]):7.1f} longest F {a[k,15]/100:5.1f}"
    if k < 12 or k % 4 == 0: print(f"{k:3d} | {cs} || {fs}{ls}")
print("chain total", us(a[nb - 2, 3]), "us; per step", us(a[nb - 2, 3]) / (nb - 1))
