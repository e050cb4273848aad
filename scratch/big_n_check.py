"""One-off scaling check beyond BASELINE's sizes: one patient, N = 8192 (D = 24) and N = 6000 (ragged, D = 64): look-ahead route vs the one-workgroup route, finite results,
agreement of nlml / gradient to 1e-9 relative (the two routes share no factorisation code), and a Richardson FD probe of three gradient components."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import medgp_amd
from medgp_amd import synth
for (N, D) in ((8192, 24), (6000, 64)):
    Q, R = 5, 8
    m, t, y = synth.patient(31, 0, D, N); th = synth.theta(31, 0, 7, Q, D, R)
    res = {}
    for route in ("1", "-1"):
        os.environ["MEDGP_MULTI_CU"] = route
        ctx = medgp_amd.Context(7, Q, D, R); ctx.reserve(1, N, 1); ctx.set_patient(0, m, t, y)
        t0 = time.perf_counter(); nl, g, st = ctx.nlml_grad([0], th[None], True); dt = time.perf_counter() - t0
        t0 = time.perf_counter(); nl, g, st = ctx.nlml_grad([0], th[None], True); dt = time.perf_counter() - t0
        res[route] = (nl[0], g[0].copy(), st[0], dt)
        if route == "1":
            fd = []
            for h in (0, D + 7, ctx.H - 3):
                vals = []
                for step in (2e-3, 1e-3):
                    tp, tm = th.copy(), th.copy(); tp[h] += step; tm[h] -= step
                    a = ctx.nlml_grad([0], tp[None], False)[0][0]; b = ctx.nlml_grad([0], tm[None], False)[0][0]
                    vals.append((a - b) / (2 * step))
                fd.append((h, (4 * vals[1] - vals[0]) / 3, g[0][h]))
        ctx.close()
    a, b = res["1"], res["-1"]
    gs = np.abs(b[1]).max()
    print(f"N={N} D={D}: status {a[2]} {b[2]}  nlml {a[0]:.6f} {b[0]:.6f} rel {abs(a[0]-b[0])/abs(b[0]):.2e}  grad max rel {np.max(np.abs(a[1]-b[1])/np.maximum(np.abs(b[1]), 1e-3*gs)):.2e}  finite {np.isfinite(a[1]).all() and np.isfinite(b[1]).all()}  ms: look-ahead {1e3*a[3]:.1f}  one workgroup {1e3*b[3]:.1f}")
    for h, f, gg in fd: print(f"   FD component {h}: {f:.8e} vs gradient {gg:.8e}  rel {abs(f-gg)/max(abs(gg),1e-3*gs):.2e}")
