"""Determinism soak of the round-6 paths (not part of the suite): python scratch/soak_r6.py SECONDS
As scratch/soak_r5.py -- ragged batches on DEFAULT routing through the blocking operator, both asynchronous lanes and medgp_screen, every
result repeated bit for bit, contexts created / destroyed along the way -- plus what round 6 added: medgp_screen on its TWO LANES (chunks
of at most max_batch / 2 entries alternating between two streams) also while an asynchronous gradient lane is in flight, contexts whose
calls run as MEMORY WAVES (a budget of a few MB, fixed per case), buffers sized by medgp_reserve_plan or grown by the calls."""
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
        if c % 3 == 0: os.environ["MEDGP_MEM_BUDGET_GB"] = "0.02"      # this case always runs in waves (same cuts every time: same bits)
        else: os.environ.pop("MEDGP_MEM_BUDGET_GB", None)
        ctx = medgp_amd.Context(7, Q, D, R); ctx.reserve(P, 6000 if c % 2 else max(ns), max(P, 64))   # odd cases: buffers on demand
        if rng.random() < 0.5: ctx.reserve_plan(ns, 5)
        ctx.set_patients(np.arange(P), [synth.patient(700 + c, p, D, n) for p, n in enumerate(ns)])
        ctx.set_prior(-1, *synth.hier_gamma_prior(Q, D, R, 0.01))
        ctxs[c] = ctx
    ctx = ctxs[c]
    H = ctx.H
    th = np.stack([synth.theta(700 + c, p, 7, Q, D, R) for p in range(P)])
    mode = int(rng.integers(4))
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
    elif mode == 2:     # screening: 5 vectors on every patient (several chunks: two lanes)
        nl, st = ctx.screen(np.arange(P), th[:5])
        res, key = (nl.tobytes(), st.tobytes()), (c, "screen")
    else:               # the same screening while an asynchronous gradient lane is in flight; the lane's results are checked too
        b = dict(th=ctx.pinned((P, H), np.float64), nl=ctx.pinned((P,), np.float64), gr=ctx.pinned((P, H), np.float64), st=ctx.pinned((P,), np.int32))
        b["th"][:] = th
        lane = int(rng.integers(2))
        ctx.nlml_grad_async(lane, np.arange(P), b["th"], True, b["nl"], b["gr"], b["st"])
        nl, st = ctx.screen(np.arange(P), th[:5])
        ctx.wait(lane)
        rl = (b["nl"].tobytes(), b["gr"].tobytes(), b["st"].tobytes())
        if (c, "grad") in ref and rl != ref[(c, "grad")]:
            print("LANE-UNDER-SCREEN MISMATCH", c, "iteration", it); sys.exit(1)
        ref.setdefault((c, "grad"), rl)
        res, key = (nl.tobytes(), st.tobytes()), (c, "screen")
    if key in ref:
        if res != ref[key]:
            print("MISMATCH", key, "iteration", it, "mode", mode); sys.exit(1)
    else:
        ref[key] = res
    it += 1; nev += P
print(f"SOAK_R6_OK {it} calls, {nev} evaluations, {len(ref)} distinct keys, {time.time() - t0:.0f} s")
