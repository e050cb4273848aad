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
