"""GPU parity tests, part 2: BASELINE configs at full size against committed fixtures, the exact bench workload,
the jitter / failure semantics against the oracle, the factor boundary (train(false) + predict in the caller's order),
reference-derived Gram matrices, the packed upload.

Tolerances as tests/test_parity_gpu.py (nlml 1e-10 relative, gradient 1e-6 of max(|g_h|, 1e-3 max|g|))."""
import os

import numpy as np
import pytest
import scipy.linalg as sla

pytestmark = pytest.mark.gpu

import medgp_amd
from medgp_amd import synth
from oracle import oracle as O
from test_parity_gpu import assert_parity, make_ctx, GOLD


def test_config5_full_size_vs_fixture():
    """BASELINE config 5 at FULL size: D=64, N=4096, Q=5, R=8 (H=2954), half of the A entries exactly zero and clamped,
    hierarchical-gamma prior (mode 2).  Expected values: committed oracle outputs (tests/golden/make_golden.py::config5)."""
    g = np.load(os.path.join(GOLD, "config5_D64_N4096.npz"))
    D, N, Q, R = int(g["D"]), int(g["N"]), int(g["Q"]), int(g["R"])
    assert (D, N) == (64, 4096)
    ctx = make_ctx(7, Q, D, R, [(g["meta"], g["t"], g["y"])])
    f, ty, ex, p0, p1 = synth.hier_gamma_prior(Q, D, R, 0.01)
    ty = ty.copy()
    ty[g["clamped"]] = 0            # test-time clamp of the exactly-zero A entries (ref: c_prior.cpp:118-140)
    ctx.set_prior(0, f, ty, ex, p0, p1)
    nlml, grad, st = ctx.nlml_grad([0], g["theta"][None, :], True)
    assert st[0] == int(g["oracle_status"]) == 0
    assert_parity(nlml[0], grad[0], {"nlml": float(g["oracle_nlml"]), "grad": g["oracle_grad"]}, "config5")
    assert np.all(grad[0][g["clamped"]] == 0.0)
    n0, _, _ = ctx.nlml_grad([0], g["theta"][None, :], False)
    assert abs(n0[0] - nlml[0]) <= 1e-12 * abs(nlml[0])
    ctx.close()


def test_headline_workload_deterministic():
    """The bench workload itself: 512 patients x N=512, D=24, Q=5, R=8, hier-gamma prior, ONE call of 512 evaluations
    (more entries than CUs: the one-workgroup-per-patient factorisation k_cholinv<4,4>, several passes per step).
    Every 8th patient plus the XCD-group / pass-boundary positions -- 70 of the 512 -- against the oracle (blocked gradient
    form, which test_oracle.py ties to the per-hyper loop of the reference; the oracle calls run on a thread pool)."""
    D, N, Q, R, P, seed = 24, 512, 5, 8, 512, 2024      # bench.py's defaults
    pts, th = synth.cohort(seed, P, D, N, Q=Q, R=R)
    ctx = medgp_amd.Context(7, Q, D, R)
    ctx.reserve(P, N, P)
    ctx.set_patients(np.arange(P), pts)
    ctx.set_prior(-1, *synth.hier_gamma_prior(Q, D, R, 0.01))
    nlml, grad, st = ctx.nlml_grad(np.arange(P), th, True)
    assert np.all(st == 0) and np.all(np.isfinite(nlml)) and np.all(np.isfinite(grad))
    pr = O.Prior.hier_gamma(Q, D, R, 0.01, 0.01)
    from concurrent.futures import ThreadPoolExecutor
    sample = sorted(set(range(0, P, 8)) | {1, 63, 255, 389, 511, 257})
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:     # ctypes releases the GIL during the oracle call
        refs = list(ex.map(lambda p: O.nlml_grad(7, Q, D, R, *pts[p], th[p], prior=pr, nthreads=1), sample))
    assert len(sample) >= 64
    for p, ref in zip(sample, refs):
        assert ref["status"] == 0
        assert_parity(nlml[p], grad[p], ref, f"p{p}")
    ctx.close()


# ---- jitter / failure semantics (ref: c_inference_exact.cpp:96-111) ---------------------------------------------------
def _dup_patient(n_unique, copies, D=1, seed=3):
    rng = np.random.default_rng(seed)
    tu = np.sort(rng.uniform(0, 60, n_unique)).astype(np.float32)
    t = np.repeat(tu, copies)
    m = np.zeros(t.size, np.int32) if D == 1 else np.tile(np.arange(D, dtype=np.int32), t.size // D + 1)[:t.size]
    y = np.sin(0.3 * t).astype(np.float32)
    return m, t, y


def _theta_noise(th, D, jit_rounds):
    """theta whose noise variance is (1 + jit_rounds) x the original: what the reference's retry loop evaluates
    after `jit_rounds` additions of the noise vector (c_inference_exact.cpp:101-104)."""
    th2 = th.copy()
    th2[:D] += 0.5 * np.log1p(jit_rounds)
    return th2


@pytest.mark.parametrize("multi_cu", ["-1", "1"])
def test_jitter_regimes_vs_oracle(multi_cu, monkeypatch):
    """Exact duplicates of well separated time stamps: K is singular up to the noise and otherwise well conditioned.
    Sweeping the noise through fp64 epsilon walks through the regimes of the reference's retry loop.  K is PSD in exact
    arithmetic, so in the rounding-dominated regime WHICH attempt first succeeds is decided by rounding errors and cannot be
    pinned across two correct implementations (the deterministic check of the retry machinery is the next test); pinned here:
      * noise well above rounding: status 0 on both sides, value parity;
      * noise absorbed by the diagonal (s + noise == s in fp64) or exactly zero: every retry adds the same nothing,
        status -1 on both sides;
      * in between: -1 <= status <= 10, finite outputs iff status >= 0;
      * a healthy neighbour in the same batch is bit-identical whatever happens to the entry beside it.
    multi_cu = 1 runs the host-driven retry loop of the multi-CU factorisation (n >= 128), -1 the in-kernel loop."""
    monkeypatch.setenv("MEDGP_MULTI_CU", multi_cu)
    D, Q, R = 1, 2, 1
    tu = (30.0 * np.arange(48)).astype(np.float32)           # 30 h apart, length scales 5-6 h: unique points ~independent
    t = np.repeat(tu, 3)                                     # n = 144: three 64-blocks
    m = np.zeros(t.size, np.int32)
    y = np.sin(0.3 * t).astype(np.float32)
    good = synth.patient(9, 0, D, 144)
    th = np.array([0.0, 0.7, -0.4, np.log(1 / 12.0), np.log(1 / 15.0), np.log(1 / (2 * synth.REF_PI * 6.0)),
                   np.log(1 / (2 * synth.REF_PI * 5.0)), np.log(0.3), np.log(0.2)])
    thg = synth.theta(2, 0, 7, Q, D, R)
    ctx = make_ctx(7, Q, D, R, [(m, t, y), good], max_batch=8)
    s_scale = O.gram(7, Q, D, R, m, t, th)[0, 0] - np.exp(2 * th[0])
    seen = []
    for rel in (1e-4, 1e-6, 1e-8, 1e-13, 3e-15, 1e-15, 4e-16, 2.3e-16, 1e-17, 0.0):
        th2 = th.copy()
        th2[0] = 0.5 * np.log(rel * s_scale) if rel > 0 else -800.0
        nlml, grad, st = ctx.nlml_grad([1, 0, 1], np.stack([thg, th2, thg]), True)
        ref = O.nlml_grad(7, Q, D, R, m, t, y, th2)
        s = int(st[1])
        seen.append(s)
        assert st[0] == 0 and st[2] == 0
        assert nlml[0] == nlml[2] and np.array_equal(grad[0], grad[2])
        assert -1 <= s <= 10
        assert np.isfinite(nlml[1]) == (s >= 0) and np.all(np.isfinite(grad[1])) == (s >= 0)
        if rel >= 1e-8:
            assert s == ref["status"] == 0
            # cos(w dt) by the angle-difference identity: |dK| <= eps |w t| ~ 1e-13 s, i.e. 1e-13 / rel relative to the noise
            assert abs(nlml[1] - ref["nlml"]) <= max(1e-10, 1e-12 / rel) * abs(ref["nlml"]), (rel, nlml[1], ref["nlml"])
        if rel < 1.1e-16:
            assert s == ref["status"] == -1
    assert 0 in seen and -1 in seen
    ctx.close()


@pytest.mark.parametrize("multi_cu,fails", [("-1", 1), ("1", 1), ("-1", 3), ("1", 2), ("-1", 11), ("1", 11)])
def test_jitter_retry_deterministic(multi_cu, fails, monkeypatch):
    """The retry machinery driven BY CONSTRUCTION, not by rounding: MEDGP_DEBUG_FAIL_ATTEMPTS=k makes the library treat the
    first k factorisation attempts of every problem as failed (a test hook; 0 in production).  The reference's loop
    (c_inference_exact.cpp:99-111) then re-adds the noise vector k times: status == k, nlml == the nlml of K + (1 + k) noise
    (the oracle at sigma' = sigma sqrt(1 + k)) to full parity, gradients likewise, except that the noise gradients use the
    ORIGINAL sigma (c_inference_exact.cpp:194-202) and are therefore the oracle's divided by (1 + k); k = 11 exhausts the
    10 retries: status -1 = the reference's `return false`.  Batch of ragged patients incl. one below the n > 2 guard."""
    monkeypatch.setenv("MEDGP_MULTI_CU", multi_cu)
    monkeypatch.setenv("MEDGP_DEBUG_FAIL_ATTEMPTS", str(fails))
    D, Q, R = 3, 2, 2
    ns = [150, 2, 200, 131]
    pts = [synth.patient(12, p, D, n, interleave=(p == 3)) for p, n in enumerate(ns)]
    th = np.stack([synth.theta(12, p, 7, Q, D, R) for p in range(len(ns))])
    ctx = make_ctx(7, Q, D, R, pts)
    nlml, grad, st = ctx.nlml_grad(np.arange(len(ns)), th, True)
    nlml0, _, st0 = ctx.nlml_grad(np.arange(len(ns)), th, False)
    assert st[1] == -1 and np.isnan(nlml[1])
    for p in (0, 2, 3):
        if fails > 10:
            assert st[p] == -1 and st0[p] == -1 and np.isnan(nlml[p]) and np.all(np.isnan(grad[p]))
            continue
        assert st[p] == fails and st0[p] == fails
        ref = O.nlml_grad(7, Q, D, R, *pts[p], _theta_noise(th[p], D, fails))
        ref["grad"][:D] /= (1 + fails)
        assert_parity(nlml[p], grad[p], ref, f"p{p} fails{fails}")
        assert abs(nlml0[p] - nlml[p]) <= 1e-12 * abs(nlml[p])
    ctx.close()


def test_jitter_exhaustion_mixed_batch_multi_cu(monkeypatch):
    """Host-driven retry of the multi-CU path with all three outcomes in one batch at n >= 128: healthy (0), exhausted (-1,
    noise exactly zero on duplicates), n <= 2 (-1 from the guard), plus whatever the borderline entry does."""
    monkeypatch.setenv("MEDGP_MULTI_CU", "1")
    D, Q, R = 2, 2, 2
    m, t, y = _dup_patient(40, 4, D=2)     # n = 160, duplicates within each output
    good = synth.patient(10, 0, D, 200)
    tiny = (np.array([0, 1], np.int32), np.array([1, 2], np.float32), np.array([0, 1], np.float32))
    th = synth.theta(4, 0, 7, Q, D, R)
    th_sing = th.copy(); th_sing[:D] = -800.0
    th_edge = th.copy(); th_edge[:D] = 0.5 * np.log(3e-16)
    ctx = make_ctx(7, Q, D, R, [(m, t, y), good, tiny], max_batch=8)
    nlml, grad, st = ctx.nlml_grad([1, 0, 2, 0, 1], np.stack([th, th_sing, th, th_edge, th]), True)
    ref = O.nlml_grad(7, Q, D, R, *good, th)
    assert st[0] == st[4] == 0 and st[1] == -1 and st[2] == -1
    assert_parity(nlml[0], grad[0], ref, "healthy")
    assert nlml[4] == nlml[0] and np.array_equal(grad[4], grad[0])
    assert np.isnan(nlml[1]) and np.isnan(nlml[2])
    assert O.nlml_grad(7, Q, D, R, m, t, y, th_sing)["status"] == -1
    assert np.isfinite(nlml[3]) == (st[3] >= 0)
    ctx.close()


@pytest.mark.parametrize("ns", [[1100, 300, 1024, 70], [1216, 1216], [1030, 2, 640, 1100, 64, 900, 130, 1088]])
def test_look_ahead_small_ragged_batches_vs_oracle(ns, monkeypatch):
    """The look-ahead schedule with 2, 4 and 8 ragged entries (the batch sizes that use the parked-neighbour workgroup once a
    step has more than 256 / nbatch tasks per entry), incl. a failing entry (n = 2) and single-block entries in the same call."""
    monkeypatch.setenv("MEDGP_MULTI_CU", "1")
    D, Q, R = 3, 2, 2
    pts = [synth.patient(31, p, D, n) for p, n in enumerate(ns)]
    ths = np.stack([synth.theta(31, p, 7, Q, D, R) for p in range(len(ns))])
    ctx = make_ctx(7, Q, D, R, pts, max_batch=8)
    nlml, grad, st = ctx.nlml_grad(np.arange(len(ns)), ths, True)
    for p, n in enumerate(ns):
        ref = O.nlml_grad(7, Q, D, R, *pts[p], ths[p], nthreads=8)
        assert st[p] == ref["status"], (p, n, st[p], ref["status"])
        if st[p] >= 0:
            assert_parity(nlml[p], grad[p], ref, f"ragged look-ahead entry {p} (n={n})")
    n2, g2, s2 = ctx.nlml_grad(np.arange(len(ns)), ths, True)        # repeatable bit for bit
    assert np.array_equal(n2[st >= 0], nlml[st >= 0]) and np.array_equal(g2[st >= 0], grad[st >= 0])
    ctx.close()


# ---- the factor boundary ---------------------------------------------------------------------------------------------
def _reference_predict(kidx, Q, D, R, m, t, alpha, linv, theta, m2, t2):
    """GP_Regression::predict restated with numpy on the reference's buffers (ref: core/gp_regression.cpp:164-196):
    mean = K*^T chol_alpha (sgemv), V = chol_factor_inv K* (strmm with the LOWER triangle), var = k** - |V col|^2 + lik."""
    n = t.size
    mm = np.concatenate([m, m2]).astype(np.int32)
    tt = np.concatenate([t, t2]).astype(np.float32)
    Kall = O.gram(kidx, Q, D, R, mm, tt, theta)         # includes the noise on its diagonal
    noise = np.exp(2 * theta[mm]) if kidx == 7 else np.full(mm.size, np.exp(2 * theta[0]))
    Ks = Kall[:n, n:]
    kss = np.diag(Kall)[n:] - noise[n:]
    mean = Ks.T @ alpha.astype(np.float64)
    V = np.tril(linv.astype(np.float64)) @ Ks
    return mean, kss - (V * V).sum(0) + noise[n:]


def test_train_false_then_reference_predict():
    """main_one_test.cpp:354-399 through the INTEGRATION.md-shaped adapter: the training set is `past + appended current
    observations` (NOT grouped by output), GP_Regression::train(false) -> compute_nlml(flag_grad = false) must leave
    chol_alpha / chol_factor_inv valid and in the caller's order for the reference's unchanged predict."""
    D, Q, R = 3, 3, 2
    m, t, y = synth.patient(21, 0, D, 150, interleave=True)       # caller order: not grouped
    th = synth.theta(21, 0, 7, Q, D, R)
    ctx = make_ctx(7, Q, D, R, [(m, t, y)])
    # flag_grad = 0 alone forms no factor: get_factor must refuse instead of returning stale data
    ctx.nlml_grad([0], th[None, :], False)
    with pytest.raises(medgp_amd.MedgpError):
        ctx.get_factor(0, t.size)
    nl, _, st = ctx.nlml_grad([0], th[None, :], False, keep_factor=True)        # train(false)
    assert st[0] == 0
    ref = O.nlml_grad(7, Q, D, R, m, t, y, th, want_alpha=True, want_linv=True)   # the oracle works in the caller's order
    assert abs(nl[0] - ref["nlml"]) <= 1e-10 * abs(ref["nlml"])
    alpha, linv, beta = ctx.get_factor(0, t.size)
    np.testing.assert_allclose(alpha, ref["alpha"], rtol=2e-6, atol=1e-6 * np.abs(ref["alpha"]).max())
    np.testing.assert_allclose(linv, ref["linv"], rtol=2e-6, atol=1e-6 * np.abs(ref["linv"]).max())
    assert np.all(np.triu(linv, 1) == 0) and abs(beta - ref["beta"]) <= 1e-6 * abs(ref["beta"])
    m2 = np.array([1, 0, 2], np.int32)
    t2 = np.array([float(t[7]), 33.25, 180.0], np.float32)
    mean, var = _reference_predict(7, Q, D, R, m, t, alpha, linv, th, m2, t2)
    rp = O.fit_predict(7, Q, D, R, m, t, y, th, m2, t2)
    np.testing.assert_allclose(mean, rp["mean"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(var, rp["var"], rtol=1e-4, atol=1e-5)
    gm, gv, gs = ctx.fit_predict(0, th, m2, t2)          # the throughput route (no inverse formed)
    assert gs == 0
    np.testing.assert_allclose(gm, rp["mean"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(gv, rp["var"], rtol=1e-5, atol=1e-6)
    with pytest.raises(medgp_amd.MedgpError):
        ctx.get_factor(0, t.size)                        # fit_predict keeps no factor
    # gradient call on the same (ungrouped) patient: evaluated grouped, L^-1 still comes back in the caller's order
    nl2, g2, _ = ctx.nlml_grad([0], th[None, :], True)
    assert_parity(nl2[0], g2[0], ref)
    a2, l2, b2 = ctx.get_factor(0, t.size)
    np.testing.assert_allclose(a2, ref["alpha"], rtol=2e-6, atol=1e-6 * np.abs(ref["alpha"]).max())
    np.testing.assert_allclose(l2, ref["linv"], rtol=2e-6, atol=1e-6 * np.abs(ref["linv"]).max())
    a3, l3, _ = ctx.get_factor(0, t.size)                # asking twice is stable
    assert np.array_equal(a3, a2) and np.array_equal(l3, l2)
    ctx.close()


@pytest.mark.parametrize("name", ["fastkernel_gram_Q5_D2_R2", "fastkernel_gram_Q5_D24_R8", "fastkernel_gram_Q3_D7_R4"])
def test_hip_factor_vs_reference_derived_gram(name):
    """Rows a3-a8 pinned end to end on the DEVICE: Gram matrices assembled from the reference's own Python factors
    (fastkernel.py, numpy's pi) + noise -> scipy Cholesky; the HIP path must reproduce nlml, alpha and L^-1 of that K."""
    g = np.load(os.path.join(GOLD, name + ".npz"))
    Q, D, R, N = int(g["Q"]), int(g["D"]), int(g["R"]), int(g["N"])
    m, t, y, th = g["meta"], g["t"], g["y"], g["theta"]
    K = g["K"] + np.diag(np.exp(2 * th[m]))
    L = sla.cholesky(K, lower=True)
    yy = y.astype(np.float64)
    alpha_ref = sla.cho_solve((L, True), yy)
    nlml_ref = 0.5 * yy @ alpha_ref + np.log(np.diag(L)).sum() + 0.5 * N * np.log(2 * np.pi)
    linv_ref = sla.solve_triangular(L, np.eye(N), lower=True)
    ctx = make_ctx(7, Q, D, R, [(m, t, y)])
    ctx.set_pi(np.pi)
    nl, _, st = ctx.nlml_grad([0], th[None, :], False, keep_factor=True)
    assert st[0] == 0
    assert abs(nl[0] - nlml_ref) <= 1e-10 * abs(nlml_ref)
    alpha, linv, _ = ctx.get_factor(0, N)
    np.testing.assert_allclose(alpha, alpha_ref, rtol=2e-6, atol=1e-6 * np.abs(alpha_ref).max())
    np.testing.assert_allclose(linv, linv_ref, rtol=2e-6, atol=1e-6 * np.abs(linv_ref).max())
    ctx.close()


@pytest.mark.parametrize("kidx", [0, 8])
def test_hip_univariate_kernels_vs_reference_python(kidx):
    """kernel_index 0 (SE) and 8 (SM) on the DEVICE against Gram matrices composed exactly as the reference's Python does
    (fastkernel.compute_se_1d / compute_sm_1d, vizkernel.py:317-320, :347-354; tests/golden/make_golden.py::fastkernel_univariate)."""
    g = np.load(os.path.join(GOLD, "fastkernel_univariate.npz"))
    t = g["t"]
    N = t.size
    th, Kref, Q = (g["hyp_se"], g["K_se"], 1) if kidx == 0 else (g["hyp_sm"], g["K_sm"], int(g["Q"]))
    rng = np.random.default_rng(4)
    y = rng.normal(size=N).astype(np.float32)
    m = np.zeros(N, dtype=np.int32)
    K = Kref + np.exp(2 * th[0]) * np.eye(N)
    L = sla.cholesky(K, lower=True)
    yy = y.astype(np.float64)
    nlml_ref = 0.5 * yy @ sla.cho_solve((L, True), yy) + np.log(np.diag(L)).sum() + 0.5 * N * np.log(2 * np.pi)
    ctx = make_ctx(kidx, Q, 1, 1, [(m, t, y)])
    ctx.set_pi(np.pi)
    nl, _, st = ctx.nlml_grad([0], th[None, :], False)
    assert st[0] == 0 and abs(nl[0] - nlml_ref) <= 1e-10 * abs(nlml_ref)
    Ld, _, st = ctx.factor(0, th, N)
    np.testing.assert_allclose(Ld @ Ld.T, K, rtol=0, atol=1e-12 * np.abs(K).max())
    ctx.close()


def test_packed_upload_equals_single_uploads():
    D, Q, R = 4, 3, 2
    ns = [1, 64, 130, 17, 200, 2]
    pts = [synth.patient(81, p, D, n, interleave=(p % 2 == 1)) for p, n in enumerate(ns)]
    th = np.stack([synth.theta(81, p, 7, Q, D, R) for p in range(len(ns))])
    a = make_ctx(7, Q, D, R, pts)
    b = medgp_amd.Context(7, Q, D, R)
    b.reserve(len(ns), max(ns), len(ns))
    b.set_patients(np.arange(len(ns))[::-1].copy(), pts[::-1])       # arbitrary slot order in one call
    ra = a.nlml_grad(np.arange(len(ns)), th, True)
    rb = b.nlml_grad(np.arange(len(ns)), th, True)
    for x, z in zip(ra, rb):
        assert np.array_equal(x, z, equal_nan=True)
    assert ra[2][0] == -1 and ra[2][5] == -1 and np.all(ra[2][1:5] == 0)
    with pytest.raises(medgp_amd.MedgpError):
        b.set_patients([0, 9], pts[:2])
    a.close(); b.close()


def test_medgp_factor_caller_order_and_prefix_property():
    """medgp_factor: L (lower, fp64) and z = L^-1 y in the CALLER's order; with time-ordered observations the leading
    p x p block / first p entries are the factor / solve of the first p observations -- the property medgp_test's shared
    imputation pass rests on (ref: main_one_test.cpp:287-300 training subsets `all observations before t`)."""
    D, Q, R, N = 3, 3, 2, 150
    m, t, y = synth.patient(33, 0, D, N, interleave=True)
    o = np.argsort(t, kind="stable")
    m, t, y = m[o], t[o], y[o]                       # time order: not grouped by output
    th = synth.theta(33, 0, 7, Q, D, R)
    ctx = make_ctx(7, Q, D, R, [(m, t, y)])
    Lm, z, st = ctx.factor(0, th, N)
    assert st == 0 and np.all(np.triu(Lm, 1) == 0)
    K = O.gram(7, Q, D, R, m, t, th)
    np.testing.assert_allclose(Lm @ Lm.T, K, rtol=0, atol=1e-12 * np.abs(K).max())
    np.testing.assert_allclose(Lm @ z, y.astype(np.float64), rtol=0, atol=1e-12)
    for p in (1, 40, 97):
        Lp = np.linalg.cholesky(K[:p, :p])
        np.testing.assert_allclose(Lm[:p, :p], Lp, rtol=0, atol=1e-11 * np.abs(Lp).max())
        np.testing.assert_allclose(z[:p], np.linalg.solve(Lp, y[:p].astype(np.float64)), rtol=0, atol=1e-10)
    ctx.close()


def test_results_do_not_depend_on_batch_size():
    """The same patient and hypers evaluated alone and as every entry of a 200-entry batch: identical bits.  (Few entries spread
    k_epilogue's hyper range -- and the prior log-density sum -- over several workgroups, many entries use one; single-block
    entries are factored by the same kernel in every call.)"""
    D, N, Q, R = 24, 64, 5, 8
    m, t, y = synth.patient(77, 0, D, N)
    th = synth.theta(77, 0, 7, Q, D, R)
    P = 200
    ctx = medgp_amd.Context(7, Q, D, R)
    ctx.reserve(P, N, P)
    ctx.set_patients(np.arange(P), [(m, t, y)] * P)
    ctx.set_prior(-1, *synth.hier_gamma_prior(Q, D, R, 0.01))
    n1, g1, s1 = ctx.nlml_grad([0], th[None], True)
    nP, gP, sP = ctx.nlml_grad(np.arange(P), np.tile(th, (P, 1)), True)
    n3, g3, s3 = ctx.nlml_grad([5, 9, 11], np.tile(th, (3, 1)), True)
    assert s1[0] == 0 and np.all(sP == 0)
    assert np.all(nP == n1[0]) and np.all(gP == g1[0]) and np.all(n3 == n1[0]) and np.all(g3 == g1[0])
    ref = O.nlml_grad(7, Q, D, R, m, t, y, th, prior=O.Prior.hier_gamma(Q, D, R, 0.01))
    assert_parity(n1[0], g1[0], ref, "batch-size invariance")
    ctx.close()


def test_beyond_baseline_sizes_the_two_routes_agree(monkeypatch):
    """One patient larger than any BASELINE config (N = 6000, not a multiple of 64, D = 64: 94 block steps of the look-ahead schedule):
    the multi-CU look-ahead route and the one-workgroup route share no factorisation code and must agree far inside the parity bars;
    three gradient components are also checked against Richardson-extrapolated central differences of the device's own nlml.
    (N = 8192, D = 24 was run once by hand with the same outcome: scratch/big_n_check.py.)"""
    D, N, Q, R = 64, 6000, 5, 8
    m, t, y = synth.patient(31, 0, D, N)
    th = synth.theta(31, 0, 7, Q, D, R)
    res = {}
    for route in ("1", "-1"):
        monkeypatch.setenv("MEDGP_MULTI_CU", route)
        ctx = medgp_amd.Context(7, Q, D, R)
        ctx.reserve(1, N, 1)
        ctx.set_patient(0, m, t, y)
        nl, g, st = ctx.nlml_grad([0], th[None], True)
        assert st[0] == 0 and np.isfinite(nl[0]) and np.isfinite(g).all()
        res[route] = (nl[0], g[0].copy())
        if route == "1":
            for h in (0, D + 7, ctx.H - 3):
                d = []
                for step in (2e-3, 1e-3):
                    tp, tm = th.copy(), th.copy()
                    tp[h] += step
                    tm[h] -= step
                    d.append((ctx.nlml_grad([0], tp[None], False)[0][0] - ctx.nlml_grad([0], tm[None], False)[0][0]) / (2 * step))
                fd = (4 * d[1] - d[0]) / 3
                assert abs(fd - g[0][h]) <= 1e-6 * max(abs(g[0][h]), 1e-3 * np.abs(g[0]).max()), (h, fd, g[0][h])
        ctx.close()
    a, b = res["1"], res["-1"]
    assert abs(a[0] - b[0]) <= 1e-12 * abs(b[0])
    gs = np.abs(b[1]).max()
    assert np.max(np.abs(a[1] - b[1]) / np.maximum(np.abs(b[1]), 1e-3 * gs)) <= 1e-9


def test_config4_full_one_call_of_4096_vs_oracle():
    """BASELINE config 4 as it is named: ONE call of 4096 evaluations at N=512, D=24 (16 passes of workgroups on 256 CUs, 147 k
    k_wgrad workgroups) -- the launch shape bench.py's `config4_full` leg times.  32 entries against the oracle: the first and the
    last entry of passes of 512 workgroup slots, the positions either side of an XCD group of eight, and a few in between."""
    D, N, Q, R, P, seed = 24, 512, 5, 8, 4096, 2024
    H = synth.num_hyp(7, Q, D, R)
    nu = 64                                              # distinct patients / hyper vectors, replicated over the 4096 slots
    pts, thu = synth.cohort(seed, nu, D, N, Q=Q, R=R)
    ctx = medgp_amd.Context(7, Q, D, R)
    ctx.reserve(P, N, P)
    # entry s holds patient (s * 37) % nu and hyper vector (s * 11 + s // nu) % nu: neighbours differ, every pair occurs
    pid = (np.arange(P) * 37) % nu
    tid = (np.arange(P) * 11 + np.arange(P) // nu) % nu
    ctx.set_patients(np.arange(P), [pts[i] for i in pid])
    ctx.set_prior(-1, *synth.hier_gamma_prior(Q, D, R, 0.01))
    th = thu[tid]
    nlml, grad, st = ctx.nlml_grad(np.arange(P), th, True)
    assert np.all(st == 0) and np.all(np.isfinite(nlml)) and np.all(np.isfinite(grad))
    assert grad.shape == (P, H)
    pr = O.Prior.hier_gamma(Q, D, R, 0.01, 0.01)
    sample = sorted({0, 1, 7, 8, 255, 256, 511, 512, 513, 1023, 1024, 1535, 1536, 2047, 2048, 2049, 2559, 2560, 3071, 3072, 3583, 3584,
                     4087, 4088, 4094, 4095, 777, 1291, 1999, 2817, 3333, 3901})
    assert len(sample) == 32
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        refs = list(ex.map(lambda s: O.nlml_grad(7, Q, D, R, *pts[pid[s]], th[s], prior=pr, nthreads=1), sample))
    for s, ref in zip(sample, refs):
        assert ref["status"] == 0
        assert_parity(nlml[s], grad[s], ref, f"s{s}")
    # equal (patient, theta) pairs anywhere in the call give equal bits
    key = pid.astype(np.int64) * nu + tid
    first = {}
    for s in range(P):
        f = first.setdefault(int(key[s]), s)
        if f != s:
            assert nlml[s] == nlml[f] and np.array_equal(grad[s], grad[f]), (s, f)
    ctx.close()


@pytest.mark.parametrize("D,Q,R,ns", [(128, 2, 3, [300, 140, 9]), (256, 2, 2, [520, 256]), (256, 1, 1, [3, 70])])
def test_many_outputs_vs_oracle(D, Q, R, ns):
    """Up to the library's limit of outputs per context (MEDGP_MAX_D = 256, medgp_dev.h; the reference's D is a free key of the
    kernel parameter list, ref: kernel/c_kernel_LMC_SM.cpp:51-70): most outputs have one observation or none, the per-output
    offset tables of the slab sums and the epilogue's D-term dot products run at their full length."""
    P = len(ns)
    pts = [synth.patient(77, p, D, n) for p, n in enumerate(ns)]
    th = np.stack([synth.theta(77, p, 7, Q, D, R, sparse_frac=0.2) for p in range(P)])
    ctx = medgp_amd.Context(7, Q, D, R)
    ctx.reserve(P, max(ns), P)
    ctx.set_patients(np.arange(P), pts)
    ctx.set_prior(-1, *synth.hier_gamma_prior(Q, D, R, 0.01))
    pr = O.Prior.hier_gamma(Q, D, R, 0.01, 0.01)
    nlml, grad, st = ctx.nlml_grad(np.arange(P), th, True)
    nlml0, _, st0 = ctx.nlml_grad(np.arange(P), th, False)
    for p in range(P):
        ref = O.nlml_grad(7, Q, D, R, *pts[p], th[p], prior=pr, nthreads=4)
        assert st[p] == ref["status"] == st0[p] == 0
        assert_parity(nlml[p], grad[p], ref, f"D{D}_p{p}")
        assert abs(nlml0[p] - ref["nlml"]) <= 1e-10 * abs(ref["nlml"])
    m2 = np.array([0, D - 1, D // 2], np.int32)
    t2 = np.array([5.0, 50.0, 150.0], np.float32)
    mean, var, pst = ctx.fit_predict(0, th[0], m2, t2)
    rp = O.fit_predict(7, Q, D, R, *pts[0], th[0], m2, t2)
    assert pst == 0
    np.testing.assert_allclose(mean, rp["mean"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(var, rp["var"], rtol=1e-5, atol=1e-6)
    ctx.close()


@pytest.mark.parametrize("multi_cu", ["-1", "1"])
def test_exported_factors_have_exact_zero_triangles(multi_cu, monkeypatch):
    """The device buffers hold leftovers outside the triangle a factorisation defines (whole-row tile stores, stale chain tiles:
    see MedgpDev::Kmat); every export must mask them.  Both routes (one workgroup per patient / look-ahead), through
    medgp_factor_batch, medgp_factor and medgp_get_factor: the strict upper triangle is EXACTLY zero and L L^T = K."""
    monkeypatch.setenv("MEDGP_MULTI_CU", multi_cu)
    D, Q, R = 3, 2, 2
    ns = [200, 70, 333, 129]
    pts = [synth.patient(41, p, D, n, interleave=(p == 1)) for p, n in enumerate(ns)]
    th = np.stack([synth.theta(41, p, 7, Q, D, R) for p in range(len(ns))])
    ctx = make_ctx(7, Q, D, R, pts)
    LZ, st = ctx.factor_batch(np.arange(len(ns)), th, ns)
    Ls = [lz[0] for lz in LZ]
    assert np.all(st == 0)
    for p, (m, t, y) in enumerate(pts):
        assert np.all(np.triu(Ls[p], 1) == 0.0)
        K = O.gram(7, Q, D, R, m, t, th[p])               # (with the noise on the diagonal, caller order)
        np.testing.assert_allclose(Ls[p] @ Ls[p].T, K, rtol=0, atol=1e-11 * np.abs(K).max())
        L1, z1, s1 = ctx.factor(p, th[p], ns[p])
        assert s1 == 0 and np.all(np.triu(L1, 1) == 0.0)
    ctx.nlml_grad(np.arange(len(ns)), th, True)
    for p in range(len(ns)):
        alpha, linv, beta = ctx.get_factor(p, ns[p])
        assert np.all(np.triu(linv, 1) == 0.0) and np.all(np.isfinite(linv))
    ctx.close()


@pytest.mark.parametrize("fails", [1, 2])
def test_jitter_retry_in_a_multi_class_call(fails, monkeypatch):
    """The reference's retry loop (ref: c_inference_exact.cpp:99-111) inside a ragged call on DEFAULT routing: every size class runs
    its own chain -- look-ahead classes hand the entries whose one attempt failed to their own k_cholinv<8,4>(sel = 2) launch, the
    workgroup-per-patient classes retry in-kernel.  With the first `fails` attempts failed by construction every entry must report
    status = fails and the values of K + (1 + fails) x noise (oracle at the shifted noise hypers), whatever its class and route."""
    monkeypatch.setenv("MEDGP_DEBUG_FAIL_ATTEMPTS", str(fails))
    D, Q, R = 3, 2, 2
    ns = [700, 300, 130, 64, 40, 1000, 5, 210, 520]
    pts = [synth.patient(55, p, D, n) for p, n in enumerate(ns)]
    th = np.stack([synth.theta(55, p, 7, Q, D, R) for p in range(len(ns))])
    ctx = make_ctx(7, Q, D, R, pts)
    nlml, grad, st = ctx.nlml_grad(np.arange(len(ns)), th, True)
    plan = ctx.last_plan()
    assert len(plan) >= 4 and any(r == 2 for _, _, r in plan) and any(r != 2 for _, _, r in plan), plan
    assert np.all(st == fails), st
    for p, (m, t, y) in enumerate(pts):
        ref = O.nlml_grad(7, Q, D, R, m, t, y, _theta_noise(th[p], D, fails), nthreads=4)
        assert ref["status"] == 0
        ref["grad"][:D] /= (1 + fails)          # the noise gradients use the ORIGINAL sigma (ref: c_inference_exact.cpp:194-202)
        assert_parity(nlml[p], grad[p], ref, f"p{p} fails{fails}")
    ctx.close()
