"""five calls of the 300-patient heavy-tailed cohort on default routing (for `rocprofv3 --kernel-trace -- python3 scratch/ragged_one.py`)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import medgp_amd
from medgp_amd import synth
P, D, Q, R = 300, 24, 5, 8
pts, th, ns = synth.ragged_cohort(0, P, D, 7, Q, R)
ctx = medgp_amd.Context(7, Q, D, R); ctx.reserve(P, int(ns.max()), P)
ctx.set_patients(np.arange(P), pts); ctx.set_prior(-1, *synth.hier_gamma_prior(Q, D, R, 0.01))
sl = np.arange(P)
for _ in range(2): ctx.nlml_grad(sl, th, True)
t0 = time.perf_counter()
for _ in range(5): ctx.nlml_grad(sl, th, True)
print("ms per call", (time.perf_counter() - t0) / 5 * 1e3, flush=True)
ctx.close()
