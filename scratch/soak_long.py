"""Long determinism soak (not part of the suite): python scratch/soak_long.py SECONDS
Random ragged batches on both factorisation routes, re-evaluated in random order for the given time; every result must repeat bit for bit
(nlml, gradient, status) and contexts are created / destroyed along the way."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import medgp_amd
from medgp_amd import synth
T = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(12345)
cases = []
for c in range(14):
    D = int(rng.choice([2, 8, 24])); Q = int(rng.choice([2, 5, 8])); R = int(min(D, rng.choice([2, 8])))
    P = int(rng.integers(1, 9)); ns = [int(rng.integers(3, 1300 if c % 3 == 0 else 400)) for _ in range(P)]
    cases.append((c, D, Q, R, ns))
ref = {}
ctxs = {}
t0 = time.time(); it = 0; nev = 0
while time.time() - t0 < T:
    c, D, Q, R, ns = cases[int(rng.integers(len(cases)))]
    route = "1" if (max(ns) > 128 and rng.random() < 0.5) else "-1"
    key = (c, route)
    os.environ["MEDGP_MULTI_CU"] = route
    if key not in ctxs or rng.random() < 0.15:
        if key in ctxs: ctxs.pop(key).close()
        ctx = medgp_amd.Context(7, Q, D, R); ctx.reserve(len(ns), max(ns), len(ns))
        ctx.set_patients(np.arange(len(ns)), [synth.patient(900 + c, p, D, n) for p, n in enumerate(ns)])
        ctx.set_prior(-1, *synth.hier_gamma_prior(Q, D, R, 0.01))
        ctxs[key] = ctx
    ctx = ctxs[key]
    th = np.stack([synth.theta(900 + c, p, 7, Q, D, R) for p in range(len(ns))])
    order = rng.permutation(len(ns))
    nl, g, st = ctx.nlml_grad(order, th[order], True)
    inv = np.argsort(order)
    res = (nl[inv].tobytes(), g[inv].tobytes(), st[inv].tobytes())
    if key in ref:
        if res != ref[key]:
            print("MISMATCH", key, ns, "iteration", it); sys.exit(1)
    else:
        ref[key] = res
        assert np.all(st >= 0), (key, st)
    it += 1; nev += len(ns)
print(f"SOAK_OK {it} calls, {nev} evaluations, {len(ref)} distinct (case, route) keys, {time.time() - t0:.0f} s")
