"""python scratch/dump_eval.py P N D out.npz  -- nlml / gradient / status of one call (LIB=path selects another build); for bit comparisons"""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import medgp_amd
from medgp_amd import capi, synth
if os.environ.get('LIB'): capi.lib_path = lambda: os.environ['LIB']
P, N, D = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
Q, R = 5, min(8, D)
ns = [N - 37 * (s % 3) for s in range(P)]          # ragged
pts = [synth.patient(21, s, D, ns[s]) for s in range(P)]
th = np.stack([synth.theta(21, s, 7, Q, D, R) for s in range(P)])
ctx = medgp_amd.Context(7, Q, D, R); ctx.reserve(P, N, P); ctx.set_patients(np.arange(P), pts)
nl, g, st = ctx.nlml_grad(np.arange(P), th, True)
np.savez(sys.argv[4], nl=nl, g=g, st=st)
print(sys.argv[4], nl[:2], st[:4])
