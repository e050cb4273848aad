"""absolute role end times per look-ahead step (lib built with -DLA_STAMPS -DLA_STAMPS_ABS -DMEDGP_STAMPS -DMEDGP_NO_DSTAMPS)"""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['MEDGP_LIB'] = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'lib_lastamps_abs.so')
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
for it in range(3):
    lib.medgp_debug_clear_slab(ctx._h, 0, 8 * 80 * 8)
    nl, g, st = ctx.nlml_grad(np.zeros(1, np.int32), th[None], True)
    buf = np.zeros(8 * 80, np.uint64)
    lib.medgp_debug_read_slab(ctx._h, 0, buf.ctypes.data_as(C.c_void_p), buf.nbytes)
a = buf.reshape(80, 8)
prev_end = None
for k in range(N // 64):
    start = int(~a[k, 7] & np.uint64(0xFFFFFFFFFFFFFFFF))
    ends = [int(a[k, r]) for r in range(3)]
    last = max(ends)
    if k in (0, 1, 2, 4, 8, 9, 16, 17, 32, 33, 48, 60) and start:
        print(f"        D own start {(int(a[k,5])-start)/100:5.1f} us after the first WG; D start -> factor done: {(int(a[k,6]))/1e3:6.1f} k s_memtime ticks")
        print(f"step {k:2d}: first WG start -> last end of D {(ends[0]-start)/100:6.1f} us  F {(ends[1]-start)/100 if ends[1] else 0:6.1f} us  L {(ends[2]-start)/100 if ends[2] else 0:6.1f} us;  gap since previous step's last end {((start-prev_end)/100 if prev_end else 0):6.1f} us")
    prev_end = last
