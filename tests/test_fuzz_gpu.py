"""GPU parity sweep over seeded random shapes: the HIP path (through the C ABI) against the CPU oracle on the corners the fixed
shapes of test_parity_gpu.py do not reach -- outputs without any observation, an output whose observations all share one time
stamp, a single component / rank one / more rank than the reference would configure, caller order shuffled, n from the
reference's minimum (3, ref: util/c_objective_one.cpp:51) up to a few 64-blocks with ragged batches, on BOTH factorisation
routes (one workgroup per patient and the multi-CU look-ahead schedule), on the library's default routing (size classes with mixed
routes) and on the nlml-only path.

Tolerances as in test_parity_gpu.py (north_star: <= 1e-6 relative on log-lik and gradients).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import medgp_amd
from medgp_amd import synth
from oracle import oracle as O

NLML_RTOL = 1e-10
GRAD_RTOL = 1e-6


def random_patient(g, D, n, mode):
    """(meta, t, y) in caller order. mode: 'plain' | 'missing' (some outputs never observed) | 'same_time' (one output observed
    only at ONE time stamp, several times) | 'shuffled' (not grouped by output) | 'burst' (all of it inside one hour)."""
    outs = np.arange(D)
    if mode == "missing" and D > 1:
        outs = np.sort(g.choice(D, size=max(1, D // 2), replace=False))
    m = np.sort(g.choice(outs, size=n)).astype(np.int32)
    span = 1.0 if mode == "burst" else 200.0
    t = g.uniform(0.0, span, size=n).astype(np.float32)
    if mode == "same_time" and n >= 6:
        d0 = m[0]
        sel = np.where(m == d0)[0][:3]
        t[sel] = t[sel[0]]
    for d in outs:                                   # the loader's order: sorted by time inside an output
        idx = np.where(m == d)[0]
        t[idx] = np.sort(t[idx])
    y = g.standard_normal(n).astype(np.float32)
    if mode == "shuffled":
        p = g.permutation(n)
        m, t, y = m[p], t[p], y[p]
    return m, t, y


CASES = []
_g = np.random.Generator(np.random.Philox(key=[20261003, 7]))
for c in range(24):
    D = int(_g.choice([1, 2, 3, 5, 8, 24, 24]))
    Q = int(_g.choice([1, 2, 3, 5, 8, 9, 12]))    # 9, 12: the two-launch split of the pair kernels (Q > 8)
    R = int(_g.choice([1, 2, min(D, 4), D]))
    P = int(_g.integers(1, 6))
    big = c % 3 == 2
    ns = [int(_g.integers(3, 330 if big else 150)) for _ in range(P)]
    mode = ["plain", "missing", "same_time", "shuffled", "burst", "plain"][c % 6]
    CASES.append((c, D, Q, R, ns, mode))


@pytest.mark.parametrize("route", ["wg", "la", "auto"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: f"c{c[0]}_D{c[1]}Q{c[2]}R{c[3]}_{c[5]}_n{'-'.join(map(str, c[4]))}")
def test_random_shapes_vs_oracle(case, route, monkeypatch):
    c, D, Q, R, ns, mode = case
    if route == "la":
        if max(ns) <= 128:
            pytest.skip("the look-ahead schedule needs at least three 64-blocks")
        monkeypatch.setenv("MEDGP_MULTI_CU", "1")
    elif route == "auto":      # default routing: size classes, each with its own route, on separate streams (round 5)
        monkeypatch.delenv("MEDGP_MULTI_CU", raising=False)
    else:
        monkeypatch.setenv("MEDGP_MULTI_CU", "-1")
    g = np.random.Generator(np.random.Philox(key=[991, c]))
    pts = [random_patient(g, D, n, mode) for n in ns]
    th = np.stack([synth.theta(991, 100 * c + p, 7, Q, D, R, sparse_frac=0.3 if c % 2 else 0.0) for p in range(len(ns))])
    ctx = medgp_amd.Context(7, Q, D, R)
    ctx.reserve(len(ns), max(ns), len(ns))
    for s, (m, t, y) in enumerate(pts):
        ctx.set_patient(s, m, t, y)
    prior = None
    if c % 3 == 0:
        ctx.set_prior(-1, *synth.hier_gamma_prior(Q, D, R, 0.01))
        prior = O.Prior.hier_gamma(Q, D, R, 0.01, 0.01)
    slots = np.arange(len(ns))
    nlml, grad, st = ctx.nlml_grad(slots, th, True)
    nlml0, _, st0 = ctx.nlml_grad(slots, th, False)
    for p, (m, t, y) in enumerate(pts):
        ref = O.nlml_grad(7, Q, D, R, m, t, y, th[p], prior=prior, nthreads=4)
        assert st[p] == ref["status"] and st0[p] == ref["status"], (p, st[p], st0[p], ref["status"])
        if ref["status"] != 0:
            continue
        assert abs(nlml[p] - ref["nlml"]) <= NLML_RTOL * abs(ref["nlml"]), (p, nlml[p], ref["nlml"])
        assert abs(nlml0[p] - ref["nlml"]) <= NLML_RTOL * abs(ref["nlml"]), (p, nlml0[p], ref["nlml"])
        gs = np.abs(ref["grad"]).max()
        err = np.abs(grad[p] - ref["grad"]) / np.maximum(np.abs(ref["grad"]), 1e-3 * gs)
        assert err.max() <= GRAD_RTOL, (p, int(err.argmax()), float(err.max()))
    ctx.close()


@pytest.mark.parametrize("kidx,Q,seed", [(0, 1, 1), (0, 1, 2), (8, 1, 3), (8, 3, 4), (8, 8, 5), (8, 12, 6)])
def test_random_single_output_families_and_predictions_vs_oracle(kidx, Q, seed):
    """The SE / SM families (kernel_index 0 / 8, ref: kernel/c_kernel_SE.cpp, c_kernel_SM.cpp) on ragged random batches, then
    fit + predict at random test times (ref: core/gp_regression.cpp:128-214) against the oracle."""
    g = np.random.Generator(np.random.Philox(key=[4242, seed]))
    ns = [int(g.integers(3, 260)) for _ in range(4)]
    pts = []
    for n in ns:
        t = np.sort(g.uniform(0.0, 150.0, size=n)).astype(np.float32)
        if n > 8:
            t[3] = t[2]                      # a repeated time stamp
        pts.append((None, t, g.standard_normal(n).astype(np.float32)))
    th = np.stack([synth.theta(4242, 10 * seed + p, kidx, Q, 1, 0) for p in range(len(ns))])
    ctx = medgp_amd.Context(kidx, Q, 1, 0)
    ctx.reserve(len(ns), max(ns), len(ns))
    for s, (_, t, y) in enumerate(pts):
        ctx.set_patient(s, None, t, y)
    nlml, grad, st = ctx.nlml_grad(np.arange(len(ns)), th, True)
    for p, (_, t, y) in enumerate(pts):
        ref = O.nlml_grad(kidx, Q, 1, 0, None, t, y, th[p])
        assert st[p] == ref["status"] == 0
        assert abs(nlml[p] - ref["nlml"]) <= NLML_RTOL * abs(ref["nlml"])
        gs = np.abs(ref["grad"]).max()
        err = np.abs(grad[p] - ref["grad"]) / np.maximum(np.abs(ref["grad"]), 1e-3 * gs)
        assert err.max() <= GRAD_RTOL, (p, int(err.argmax()), float(err.max()))
        ts = g.uniform(-5.0, 160.0, size=7).astype(np.float32)
        mean, var, pst = ctx.fit_predict(p, th[p], None, ts)
        assert pst == 0
        rp = O.fit_predict(kidx, Q, 1, 0, None, t, y, th[p], None, ts)
        np.testing.assert_allclose(mean, rp["mean"], rtol=2e-5, atol=2e-6)     # float outputs (ref: vector<float>)
        np.testing.assert_allclose(var, rp["var"], rtol=2e-5, atol=2e-6)
    ctx.close()
