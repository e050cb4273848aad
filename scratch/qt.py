"""quick timing: python scratch/qt.py P N D [flag_grad] -- per-kernel HIP-event ms (LIB=path selects another build)"""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import medgp_amd
from medgp_amd import capi, synth
if os.environ.get('LIB'): capi.lib_path = lambda: os.environ['LIB']
P, N, D = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
fg = int(sys.argv[4]) if len(sys.argv) > 4 else 1
Q, R = 5, min(8, D)
nu = min(P, 8)
pts, th = synth.cohort(11, nu, D, N, Q=Q, R=R)
ctx = medgp_amd.Context(7, Q, D, R); ctx.reserve(P, N + int(os.environ.get("QT_PAD", "0")), P)
ctx.set_patients(np.arange(P), [pts[s % nu] for s in range(P)])
th = np.stack([th[s % nu] for s in range(P)])
nl, g, st = ctx.nlml_grad(np.arange(P), th, bool(fg))
import time
for _ in range(3): ctx.nlml_grad(np.arange(P), th, bool(fg))
t0 = time.perf_counter()
for _ in range(10): ctx.nlml_grad(np.arange(P), th, bool(fg))
wall = (time.perf_counter() - t0) / 10 * 1e3
ctx.profile_enable(True)
reps = 5
for _ in range(reps): ctx.nlml_grad(np.arange(P), th, bool(fg))
prof = {k: round(v[0] / reps, 3) for k, v in ctx.profile_read().items() if v[1] > 0}
print(os.environ.get('LIB', 'default'), f"P{P} N{N} D{D} fg{fg}", 'nlml0', repr(nl[0]), 'st', st[:2], prof, 'sum', round(sum(prof.values()), 3), 'wall_ms_per_call', round(wall, 3), flush=True)
