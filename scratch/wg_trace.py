"""k_wgrad placement / timing trace (library built with -DMEDGP_STAMPS -DMEDGP_NO_DSTAMPS -DMEDGP_WG_TRACE): python scratch/wg_trace.py N D"""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['MEDGP_LIB'] = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'lib_wgtrace.so')
import medgp_amd
from medgp_amd import capi, synth
N, D = int(sys.argv[1]), int(sys.argv[2])
Q, R = 5, 8
m, t, y = synth.patient(11, 0, D, N); th = synth.theta(11, 0, 7, Q, D, R)
ctx = medgp_amd.Context(7, Q, D, R); ctx.reserve(1, N, 1); ctx.set_patient(0, m, t, y)
lib = capi.load()
nl = np.empty(1); g = np.empty((1, ctx.H)); st = np.empty(1, np.int32); sl = np.zeros(1, np.int32)
lib.medgp_debug_read_xk.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int]
nb = N // 64; ntiles = nb * (nb + 1) // 2
for it in range(3):
    buf = np.zeros(3 * 1300, np.uint64)
    lib.medgp_debug_read_xk(ctx._h, 0, buf.ctypes.data_as(C.c_void_p), buf.nbytes, 1)
    lib.medgp_nlml_grad(ctx._h, 1, sl.ctypes.data_as(C.POINTER(C.c_int32)), th.ctypes.data_as(C.POINTER(C.c_double)), 1, nl.ctypes.data_as(C.POINTER(C.c_double)), g.ctypes.data_as(C.POINTER(C.c_double)), st.ctypes.data_as(C.POINTER(C.c_int32)))
    lib.medgp_debug_read_xk(ctx._h, 0, buf.ctypes.data_as(C.c_void_p), buf.nbytes, 0)
r = buf.reshape(-1, 3)[:min(ntiles, 1300)]
hw = r[:, 0]; xcc = (hw >> 32) & 0xf; cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
key = xcc * 1000 + se * 100 + sh * 10 * 0 + cu + sh * 16
t0 = r[2:, 1].min()
start = (r[:, 1].astype(np.int64) - int(t0)) / 100.0; end = (r[:, 2].astype(np.int64) - int(t0)) / 100.0   # us
def length(x):
    i = int((np.sqrt(8.0 * x + 1) - 1) / 2)
    while (i + 1) * (i + 2) // 2 <= x: i += 1
    while i * (i + 1) // 2 > x: i -= 1
    return nb - i
print("kernel span %.1f us, %d tiles; distinct CUs %d" % (end[2:].max(), ntiles, len(set(key[2:].tolist()))))
for x in (2, 3, 4, 8, 16, 64, 128, 255, 256, 257, 300, 400, 500, 527):
    if x < len(r): print(f"wg {x:4d} len {length(x):2d} xcc {xcc[x]} se {se[x]} sh {sh[x]} cu {cu[x]:2d}  start {start[x]:7.1f} end {end[x]:7.1f}  dur {end[x]-start[x]:6.1f} us")
# per-CU load
from collections import defaultdict
d = defaultdict(list)
for x in range(2, len(r)): d[int(key[x])].append(x)
loads = sorted(((max(end[x] for x in v), sum(length(x) for x in v), v) for v in d.values()), reverse=True)
for e, L, v in loads[:6]: print("CU last end %.1f us, units %d, wgs %s" % (e, L, v))
for e, L, v in loads[-3:]: print("CU last end %.1f us, units %d, wgs %s" % (e, L, v))
dur = end - start
for Lx in (32, 24, 16, 8, 4, 1):
    xs = [x for x in range(2, len(r)) if length(x) == Lx]
    if xs: print(f"len {Lx:2d}: mean dur {np.mean(dur[xs]):6.1f} us  ({len(xs)} tiles)  us/unit {np.mean(dur[xs])/Lx:.2f}")
