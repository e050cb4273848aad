"""bench-like kernel timing on the bench workload (512 distinct patients), no result checks: python scratch/qb.py [P N]"""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import medgp_amd
from medgp_amd import synth
P = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
D, Q, R = 24, 5, 8
pts = [synth.patient(2024, s, D, N) for s in range(P)]
th = np.stack([synth.theta(2024, s, 7, Q, D, R) for s in range(P)])
ctx = medgp_amd.Context(7, Q, D, R); ctx.reserve(P, N, P); ctx.set_patients(np.arange(P), pts)
ctx.set_prior(-1, *synth.hier_gamma_prior(Q, D, R, 0.01))
for _ in range(3): ctx.nlml_grad(np.arange(P), th, True)
ctx.profile_enable(True)
for _ in range(8): ctx.nlml_grad(np.arange(P), th, True)
print(os.environ.get('MEDGP_LIB', 'default'), {k: round(v[0] / 8, 4) for k, v in ctx.profile_read().items() if v[1] > 0})
