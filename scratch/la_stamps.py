import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['MEDGP_LIB'] = os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ.get('LASTAMP_LIB', 'lib_lastamps.so'))
os.environ['MEDGP_DBG_NOWGRAD'] = '1'
import medgp_amd
from medgp_amd import capi, synth
N, D = int(sys.argv[1]), int(sys.argv[2])
Q, R = 5, 8
m, t, y = synth.patient(11, 0, D, N); th = synth.theta(11, 0, 7, Q, D, R)
ctx = medgp_amd.Context(7, Q, D, R); ctx.reserve(1, N, 1); ctx.set_patient(0, m, t, y)
lib = capi.load()
nl = np.empty(1); g = np.empty((1, ctx.H)); st = np.empty(1, np.int32); sl = np.zeros(1, np.int32)
for it in range(2):
    buf = np.zeros(2048 + 16 * 80, np.uint64)
    lib.medgp_nlml_grad(ctx._h, 1, sl.ctypes.data_as(C.POINTER(C.c_int32)), th.ctypes.data_as(C.POINTER(C.c_double)), 1, nl.ctypes.data_as(C.POINTER(C.c_double)), g.ctypes.data_as(C.POINTER(C.c_double)), st.ctypes.data_as(C.POINTER(C.c_int32)))
    lib.medgp_debug_read_slab.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    lib.medgp_debug_read_slab(ctx._h, 0, buf.ctypes.data_as(C.c_void_p), buf.nbytes)
f = buf[640:640 + 16 * 80].reshape(80, 16).astype(np.float64) / 1000.0
a = buf[:640].reshape(80, 8).astype(np.float64) / 1000.0   # s_memtime ticks = shader cycles -> kilo-cycles
for k in (0, 1, 2, 3, 4, 5, 8, 16, 31, 32, 48, 62):
    if k < N // 64: print(f"   D phases (kcyc since launch): loads+gemm {a[k,3]:.1f}  trsm {a[k,4]:.1f}  update {a[k,5]:.1f}  factor {a[k,6]:.1f}  end {a[k,0]:.1f}"); print("   F key row (kcyc): status %.1f  init+partials %.1f  gemm(k-1) %.1f  trsm Ls %.1f  own trsm+rank64+store %.1f  self product %.1f  gemm2+dterm %.1f | next M row: %.1f %.1f %.1f %.1f %.1f" % (tuple(f[k, :7]) + tuple(f[k, 8:13]))); print(f"step {k:2d}: longest D {a[k,0]:7.1f} kcyc   F {a[k,1]:7.1f} kcyc   L {a[k,2]:7.1f} kcyc")
# cumulative wave-0 stamps inside diag_factor_wg (-DLA_FSTAMPS): d16(0) | A0 | trail0 | d16(1) | A1 | trail1 | d16(2) | A2 | trail2 | d16(3) | loop end | tail
fs = buf[2048:].reshape(80, 16).astype(np.float64) / 1000.0
for k in (1, 4, 16, 30):
    if k < N // 64 - 1: print(f"step {k:2d} factor (kcyc, cumulative, wave 0): " + " ".join(f"{v:.2f}" for v in fs[k, :12]))
