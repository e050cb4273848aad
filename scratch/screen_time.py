"""kernel times of the nlml-only random-init screening batch (1000 hyper vectors of one N=512 patient; ref: main_one_train.cpp:228-253)
and of the headline step, no result checks (used with diagnostic libraries whose results are meaningless): python scratch/screen_time.py"""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import medgp_amd
from medgp_amd import synth
D, N, Q, R, P = 24, 512, 5, 8, 1000
ctx = medgp_amd.Context(7, Q, D, R); ctx.reserve(1, N, P)
ctx.set_patient(0, *synth.patient(2026, 0, D, N))
th = np.stack([synth.theta(2026, s, 7, Q, D, R) for s in range(P)])
slots = np.zeros(P, dtype=np.int32)
for _ in range(3): ctx.nlml_grad(slots, th, False)
ctx.profile_enable(True)
for _ in range(8): ctx.nlml_grad(slots, th, False)
print(os.environ.get('MEDGP_LIB', 'default'), 'screening 1000 x N=512 nlml-only', {k: round(v[0] / 8, 4) for k, v in ctx.profile_read().items() if v[1] > 0})
