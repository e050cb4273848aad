"""Soak: random shapes, ragged n; (1) bitwise run-to-run reproducibility of nlml / gradients (race detector),
(2) parity of a few entries against the oracle."""
import sys, time, os, numpy as np
sys.path.insert(0, '/root/repo')
import medgp_amd
from medgp_amd import synth
from oracle import oracle as O
rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "7")))
t_end = time.time() + float(os.environ.get("SOAK_SECONDS", "150"))
it = 0; worst_n = worst_g = 0.0
while time.time() < t_end:
    kidx = int(rng.choice([7, 7, 7, 8, 0]))
    D = int(rng.choice([1, 2, 3, 6, 13, 24])); Q = int(rng.integers(1, 6)); R = int(min(D, rng.integers(1, 5)))
    if kidx != 7: D = 1
    if kidx == 0: Q = 1
    N = int(rng.choice([40, 64, 65, 130, 200, 257, 384, 512, 600, 1024]))
    P = int(rng.choice([3, 17, 64, 70, 130, 256, 300, 512]))
    if N >= 600: P = min(P, 130)
    if os.environ.get("SOAK_SINGLE") == "1": os.environ["MEDGP_MULTI_CU"] = "-1"
    pts, th = synth.cohort(int(rng.integers(1, 10**6)), P, D, N, kernel_index=kidx, Q=Q, R=R)
    ns = rng.integers(max(3, N // 3), N + 1, size=P)
    pts = [(m[:k], t[:k], y[:k]) for (m, t, y), k in zip(pts, ns)]
    ctx = medgp_amd.Context(kidx, Q, D, R); ctx.reserve(P, N, P)
    use_prior = kidx == 7 and rng.random() < 0.5
    pr = synth.hier_gamma_prior(Q, D, R, 0.01) if use_prior else None
    if pr is not None: ctx.set_prior(-1, *pr)
    for s, (m, t, y) in enumerate(pts): ctx.set_patient(s, m if kidx == 7 else None, t, y)
    a = ctx.nlml_grad(np.arange(P), th, True)
    b = ctx.nlml_grad(np.arange(P), th, True)
    c = ctx.nlml_grad(np.arange(P), th, False)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]), ("not reproducible", D, Q, R, N, P)
    assert np.array_equal(a[0], c[0]), ("nlml-only differs", D, Q, R, N, P)
    assert (a[2] >= 0).all(), ("status", D, Q, R, N, P, a[2][a[2] < 0][:5])
    for p in rng.choice(P, size=min(P, 3), replace=False):
        m, t, y = pts[p]
        ref = O.nlml_grad(kidx, Q, D, R, m, t, y, th[p], nthreads=8, prior=(O.Prior.hier_gamma(Q, D, R, beta_lam=0.01) if use_prior else None))
        en = abs(a[0][p] - ref['nlml']) / abs(ref['nlml']); gs = np.abs(ref['grad']).max()
        eg = (np.abs(a[1][p] - ref['grad']) / np.maximum(np.abs(ref['grad']), 1e-3 * gs)).max()
        worst_n = max(worst_n, en); worst_g = max(worst_g, eg)
        assert en < 1e-10 and eg < 1e-6, ("parity", kidx, D, Q, R, N, P, int(p), en, eg)
    ctx.close(); it += 1
print(f"soak ok: {it} random shapes, worst nlml rel {worst_n:.1e}, worst grad rel {worst_g:.1e}")
