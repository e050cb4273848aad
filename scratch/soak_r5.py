"""Determinism soak of the round-5 paths (not part of the suite): python scratch/soak_r5.py SECONDS
Ragged batches on DEFAULT routing (size classes with mixed routes on separate streams), evaluated in random caller order through the
blocking operator, through the two asynchronous lanes (copy streams) with both lanes in flight, and through medgp_screen; every
result must repeat bit for bit; contexts are created / destroyed along the way."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import medgp_amd
from medgp_amd import synth
T = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(2025)
cases = []
for c in range(10):
    D = int(rng.choice([2, 8, 24])); Q = int(rng.choice([2, 5])); R = int(min(D, rng.choice([2, 8])))
    P = int(rng.integers(6, 60))
    ns = [int(min(2500, max(3, np.exp(np.log(120) + 1.1 * rng.standard_normal())))) for _ in range(P)]
    cases.append((c, D, Q, R, ns))
ref, ctxs = {}, {}
t0 = time.time(); it = 0; nev = 0
while time.time() - t0 < T:
    c, D, Q, R, ns = cases[int(rng.integers(len(cases)))]
    P = len(ns)
    if c not in ctxs or rng.random() < 0.1:
        if c in ctxs: ctxs.pop(c).close()
        ctx = medgp_amd.Context(7, Q, D, R); ctx.reserve(P, max(ns), max(P, 64))
        ctx.set_patients(np.arange(P), [synth.patient(700 + c, p, D, n) for p, n in enumerate(ns)])
        ctx.set_prior(-1, *synth.hier_gamma_prior(Q, D, R, 0.01))
        ctxs[c] = ctx
    ctx = ctxs[c]
    H = ctx.H
    th = np.stack([synth.theta(700 + c, p, 7, Q, D, R) for p in range(P)])
    mode = int(rng.integers(3))
    if mode == 0:       # blocking operator, random caller order
        order = rng.permutation(P)
        nl, g, st = ctx.nlml_grad(order, th[order], True)
        inv = np.argsort(order)
        res = (nl[inv].tobytes(), g[inv].tobytes(), st[inv].tobytes())
        key = (c, "grad")
    elif mode == 1:     # both lanes in flight on the same composition (two copies of the call), then a blocking call in between
        bufs = []
        for lane in range(2):
            b = dict(th=ctx.pinned((P, H), np.float64), nl=ctx.pinned((P,), np.float64), gr=ctx.pinned((P, H), np.float64), st=ctx.pinned((P,), np.int32))
            b["th"][:] = th; b["gr"][:] = -3.0
            bufs.append(b)
            ctx.nlml_grad_async(lane, np.arange(P), b["th"], True, b["nl"], b["gr"], b["st"])
        ctx.wait(1); ctx.wait(0)
        r0 = (bufs[0]["nl"].tobytes(), bufs[0]["gr"].tobytes(), bufs[0]["st"].tobytes())
        r1 = (bufs[1]["nl"].tobytes(), bufs[1]["gr"].tobytes(), bufs[1]["st"].tobytes())
        if r0 != r1:
            print("LANE MISMATCH", c, "iteration", it); sys.exit(1)
        res, key = r0, (c, "grad")
    else:               # screening: 5 vectors on every patient, against the operator's nlml-only bits of the same composition
        nl, st = ctx.screen(np.arange(P), th[:5])
        res, key = (nl.tobytes(), st.tobytes()), (c, "screen")
    if key in ref:
        if res != ref[key]:
            print("MISMATCH", key, "iteration", it, "mode", mode); sys.exit(1)
    else:
        ref[key] = res
    it += 1; nev += P
print(f"SOAK_R5_OK {it} calls, {nev} evaluations, {len(ref)} distinct keys, {time.time() - t0:.0f} s")
