"""Short randomised soak on the GPU: bitwise run-to-run reproducibility (no atomics, fixed reduction orders -- also a
race detector for the barrier-light factorisation kernel), nlml-only == nlml+grad bits, parity of sampled entries."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import medgp_amd
from medgp_amd import synth
from oracle import oracle as O


@pytest.mark.parametrize("single_wg", [False, True])
def test_random_shapes_reproducible_and_in_parity(single_wg, monkeypatch):
    if single_wg:
        monkeypatch.setenv("MEDGP_MULTI_CU", "-1")
    rng = np.random.default_rng(99 + single_wg)
    t_end = time.time() + 12.0
    shapes = 0
    while time.time() < t_end or shapes < 6:
        D = int(rng.choice([1, 2, 3, 6, 13, 24]))
        Q = int(rng.integers(1, 6))
        R = int(min(D, rng.integers(1, 5)))
        N = int(rng.choice([40, 64, 65, 130, 200, 257, 384, 512, 600]))
        P = int(rng.choice([3, 17, 64, 70, 130, 256, 300, 512]))
        if N >= 600:
            P = min(P, 130)
        pts, th = synth.cohort(int(rng.integers(1, 10 ** 6)), P, D, N, Q=Q, R=R)
        ns = rng.integers(max(3, N // 3), N + 1, size=P)
        pts = [(m[:k], t[:k], y[:k]) for (m, t, y), k in zip(pts, ns)]
        ctx = medgp_amd.Context(7, Q, D, R)
        ctx.reserve(P, N, P)
        for s, (m, t, y) in enumerate(pts):
            ctx.set_patient(s, m, t, y)
        a = ctx.nlml_grad(np.arange(P), th, True)
        b = ctx.nlml_grad(np.arange(P), th, True)
        c = ctx.nlml_grad(np.arange(P), th, False)
        tag = (D, Q, R, N, P)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]), tag
        assert np.array_equal(a[0], c[0]) and (a[2] >= 0).all(), tag
        p = int(rng.integers(0, P))
        m, t, y = pts[p]
        ref = O.nlml_grad(7, Q, D, R, m, t, y, th[p], nthreads=4)
        gs = np.abs(ref["grad"]).max()
        assert abs(a[0][p] - ref["nlml"]) <= 1e-10 * abs(ref["nlml"]), tag
        assert (np.abs(a[1][p] - ref["grad"]) / np.maximum(np.abs(ref["grad"]), 1e-3 * gs)).max() <= 1e-6, tag
        ctx.close()
        shapes += 1


def test_round6_paths_repeat_bit_for_bit(monkeypatch):
    """Twelve seconds of the round-6 machinery under random interleaving (the five-minute version is scratch/soak_r6.py: 64 k calls,
    1.7 M evaluations, no mismatch): medgp_screen on its two lanes -- also with an asynchronous gradient lane in flight --, calls that
    run as memory waves, buffers grown by the calls or sized by medgp_reserve_plan; every result must repeat bit for bit."""
    rng = np.random.default_rng(606)
    cases = []
    for c in range(4):
        D, Q, R = [(2, 2, 2), (8, 5, 2), (24, 5, 8), (8, 2, 8)][c]
        P = int(rng.integers(8, 40))
        ns = [int(min(1500, max(3, np.exp(np.log(120) + 1.1 * rng.standard_normal())))) for _ in range(P)]
        cases.append((c, D, Q, R, ns))
    ref, ctxs = {}, {}
    t_end = time.time() + 12.0
    it = 0
    while time.time() < t_end or it < 24:
        c, D, Q, R, ns = cases[int(rng.integers(len(cases)))]
        P = len(ns)
        if c not in ctxs or rng.random() < 0.15:
            if c in ctxs:
                ctxs.pop(c).close()
            if c % 2 == 0:
                monkeypatch.setenv("MEDGP_MEM_BUDGET_GB", "0.02")     # this case always runs in waves
            else:
                monkeypatch.delenv("MEDGP_MEM_BUDGET_GB", raising=False)
            ctx = medgp_amd.Context(7, Q, D, R)
            ctx.reserve(P, 6000 if c >= 2 else max(ns), 16)            # max_batch 16 < 5 P entries: several screening chunks, two lanes
            ctx.set_patients(np.arange(P), [synth.patient(900 + c, p, D, n) for p, n in enumerate(ns)])
            ctx.set_prior(-1, *synth.hier_gamma_prior(Q, D, R, 0.01))
            if rng.random() < 0.5:
                ctx.reserve_plan(ns, 5)
            ctxs[c] = ctx
        ctx = ctxs[c]
        H = ctx.H
        th = np.stack([synth.theta(900 + c, p, 7, Q, D, R) for p in range(P)])
        sel = np.sort(rng.choice(P, size=min(P, 16), replace=False))
        mode = int(rng.integers(3))
        if mode == 0:
            nl, g, st = ctx.nlml_grad(sel, th[sel], True)
            res, key = (nl.tobytes(), g.tobytes(), st.tobytes()), (c, "grad", sel.tobytes())
        elif mode == 1:
            nl, st = ctx.screen(np.arange(P), th[:5])
            res, key = (nl.tobytes(), st.tobytes()), (c, "screen")
        else:
            b = dict(th=ctx.pinned((len(sel), H), np.float64), nl=ctx.pinned((len(sel),), np.float64), gr=ctx.pinned((len(sel), H), np.float64),
                     st=ctx.pinned((len(sel),), np.int32))
            b["th"][:] = th[sel]
            lane = int(rng.integers(2))
            ctx.nlml_grad_async(lane, sel, b["th"], True, b["nl"], b["gr"], b["st"])
            nl, st = ctx.screen(np.arange(P), th[:5])
            ctx.wait(lane)
            kg = (c, "grad", sel.tobytes())
            rl = (b["nl"].tobytes(), b["gr"].tobytes(), b["st"].tobytes())
            assert ref.setdefault(kg, rl) == rl, ("lane under screen", c, it)
            res, key = (nl.tobytes(), st.tobytes()), (c, "screen")
        assert ref.setdefault(key, res) == res, (key[:2], it, mode)
        it += 1
    for ctx in ctxs.values():
        ctx.close()
