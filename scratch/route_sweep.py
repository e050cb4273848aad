"""(N, nbatch) table of the factorisation routes (verdict item 5): ms of the factorisation stage (k_cholinv / k_la_*) per call for
the multi-CU look-ahead schedule and the two single-workgroup shapes; nlml + gradient calls, D = 24 (D = 2 with --d2).
usage: python scratch/route_sweep.py [--d2]"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import medgp_amd
from medgp_amd import synth
D = 2 if "--d2" in sys.argv else 24
Q, R = 5, min(8, D)
Ns = (128, 256, 384, 512, 768, 1024)
Ps = (8, 16, 32, 64, 96, 128, 160, 192, 256)
variants = (("LA", {"MEDGP_MULTI_CU": "1"}), ("WG44", {"MEDGP_MULTI_CU": "-1", "MEDGP_CHOLINV_NW": "44"}), ("WG84", {"MEDGP_MULTI_CU": "-1", "MEDGP_CHOLINV_NW": "84"}))
print(f"D={D}: factorisation ms per call (k_cholinv + k_la_step + k_la_aux), best marked *")
print("N     P   " + "  ".join(f"{v[0]:>8s}" for v in variants))
for N in Ns:
    nu = 8
    pts, th0 = synth.cohort(11, nu, D, N, Q=Q, R=R)
    for P in Ps:
        if P * N * N * 16 > 24e9: continue
        res = []
        for name, env in variants:
            for k in ("MEDGP_MULTI_CU", "MEDGP_CHOLINV_NW"): os.environ.pop(k, None)
            os.environ.update(env)
            ctx = medgp_amd.Context(7, Q, D, R); ctx.reserve(P, N, P)
            ctx.set_patients(np.arange(P), [pts[s % nu] for s in range(P)])
            th = np.stack([th0[s % nu] for s in range(P)])
            for _ in range(3): ctx.nlml_grad(np.arange(P), th, True)
            ctx.profile_enable(True)
            reps = 5
            for _ in range(reps): ctx.nlml_grad(np.arange(P), th, True)
            prof = ctx.profile_read()
            ms = sum(v[0] for k, v in prof.items() if k in ("k_cholinv", "k_la_step", "k_la_aux")) / reps
            res.append(ms)
            ctx.close()
        b = int(np.argmin(res))
        print(f"{N:5d} {P:4d} " + "  ".join(f"{r:8.3f}{'*' if i == b else ' '}" for i, r in enumerate(res)), flush=True)
