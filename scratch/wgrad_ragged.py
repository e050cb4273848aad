"""k_wgrad (HIP-event ms) of small sets of large patients of DIFFERENT sizes in one call: python scratch/wgrad_ragged.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import medgp_amd
from medgp_amd import synth
D, Q, R = 24, 5, 8
sets = [[3595, 2239], [3595, 3001, 2406, 2239], [3595, 2400, 2400, 2400], [3595, 2400, 2400, 2400, 2400], [3595] + [2400] * 7, [3595] + [2400] * 8, [3595] * 4, [2048] * 4, [2048] * 8]
allns = sorted({n for s in sets for n in s})
pts = {n: synth.patient(1, n, D, n) for n in allns}
for ns in sets:
    ctx = medgp_amd.Context(7, Q, D, R); ctx.reserve(len(ns), max(ns), len(ns))
    ctx.set_patients(np.arange(len(ns)), [pts[n] for n in ns])
    th = np.stack([synth.theta(1, p, 7, Q, D, R) for p in range(len(ns))])
    sl = np.arange(len(ns))
    for _ in range(2): ctx.nlml_grad(sl, th, True)
    ctx.profile_reset(); ctx.profile_enable(True)
    for _ in range(3): ctx.nlml_grad(sl, th, True)
    pr = {k: round(v[0] / 3, 3) for k, v in ctx.profile_read().items() if v[1] > 0}
    fl = sum(float(n) ** 3 / 3 for n in ns)
    print(ns, "k_wgrad", pr.get("k_wgrad"), "ms =", round(fl / pr["k_wgrad"] / 1e9, 1), "TFLOP/s; k_la_step", pr.get("k_la_step"), "plan", ctx.last_plan(), flush=True)
    ctx.close()
