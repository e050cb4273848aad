"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle, the committed golden
fixtures, and size-independent properties at BASELINE.json's full sizes.

Tolerances (SURVEY section 8c contract 1, north_star "<= 1e-6 rel on log-lik and gradients"):
  nlml  : |d| <= 1e-10 * |ref|
  grad_h: |d| <= 1e-6 * max(|ref_h|, 1e-3 * max|ref|)      (observed ~1e-11)
"""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import medgp_amd
from medgp_amd import synth
from oracle import oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")
NLML_RTOL = 1e-10
GRAD_RTOL = 1e-6


def assert_parity(nlml, grad, ref, tag=""):
    assert abs(nlml - ref["nlml"]) <= NLML_RTOL * abs(ref["nlml"]), (tag, nlml, ref["nlml"])
    if grad is not None:
        gs = np.abs(ref["grad"]).max()
        err = np.abs(grad - ref["grad"]) / np.maximum(np.abs(ref["grad"]), 1e-3 * gs)
        assert err.max() <= GRAD_RTOL, (tag, int(err.argmax()), err.max())


def make_ctx(kidx, Q, D, R, pts, max_batch=None):
    ctx = medgp_amd.Context(kidx, Q, D, R)
    nmax = max(p[1].shape[0] for p in pts)
    ctx.reserve(len(pts), max(nmax, 1), max_batch or len(pts))
    for s, (m, t, y) in enumerate(pts):
        ctx.set_patient(s, m if kidx == 7 else None, t, y)
    return ctx


@pytest.mark.parametrize("D,N,Q,R,P,interleave", [
    (2, 150, 5, 2, 3, False),     # BASELINE config 1 shape
    (2, 256, 5, 2, 8, False),     # config 2 shape (reduced P)
    (2, 97, 5, 2, 3, True),       # ragged n, caller order not grouped by output
    (24, 512, 5, 8, 2, False),    # config 4 shape (reduced P)
    (7, 65, 3, 4, 4, True),
    (3, 3, 2, 2, 2, False),       # smallest n the reference accepts
])
def test_lmc_parity_vs_oracle(D, N, Q, R, P, interleave):
    pts, th = synth.cohort(101, P, D, N, Q=Q, R=R, interleave=interleave)
    ctx = make_ctx(7, Q, D, R, pts)
    nlml, grad, st = ctx.nlml_grad(np.arange(P), th, True)
    nlml0, _, st0 = ctx.nlml_grad(np.arange(P), th, False)
    for p, (m, t, y) in enumerate(pts):
        ref = O.nlml_grad(7, Q, D, R, m, t, y, th[p], nthreads=4)
        assert st[p] == ref["status"] == 0 and st0[p] == 0
        assert_parity(nlml[p], grad[p], ref, f"p{p}")
        assert nlml0[p] == nlml[p]    # nlml-only path: same bits
    ctx.close()


def test_ragged_batch_and_prior():
    D, Q, R = 4, 3, 2
    ns = [5, 64, 65, 130, 17, 200]
    pts = [synth.patient(55, p, D, n) for p, n in enumerate(ns)]
    th = np.stack([synth.theta(55, p, 7, Q, D, R, sparse_frac=0.5) for p in range(len(ns))])
    ctx = make_ctx(7, Q, D, R, pts)
    f, ty, ex, p0, p1 = synth.hier_gamma_prior(Q, D, R, 0.01)
    # test-time clamp of the exactly-zero A entries of patient 2 (ref: c_prior.cpp:118-140)
    ty2, f2 = ty.copy(), f.copy()
    z = np.where(th[2][D:D + Q * D * R] == 0.0)[0] + D
    ty2[z] = 0
    ctx.set_prior(-1, f, ty, ex, p0, p1)
    ctx.set_prior(2, f2, ty2, ex, p0, p1)
    ctx.set_prior(4)   # no prior on patient 4
    nlml, grad, st = ctx.nlml_grad(np.arange(len(ns)), th, True)
    for p, (m, t, y) in enumerate(pts):
        pr = O.Prior.hier_gamma(Q, D, R, 0.01, 0.01)
        if p == 2:
            pr.type[z] = 0
        if p == 4:
            pr = None
        ref = O.nlml_grad(7, Q, D, R, m, t, y, th[p], prior=pr)
        assert_parity(nlml[p], grad[p], ref, f"p{p}")
        if p == 2:
            assert np.all(grad[p][z] == 0.0)
    ctx.close()


@pytest.mark.parametrize("kidx,Q", [(8, 4), (0, 1)])
def test_single_output_kernels(kidx, Q):
    pts = [synth.patient(77, p, 1, 120 + 7 * p) for p in range(3)]
    th = np.stack([synth.theta(77, p, kidx, Q, 1, 0) for p in range(3)])
    ctx = make_ctx(kidx, Q, 1, 0, pts)
    nlml, grad, st = ctx.nlml_grad(np.arange(3), th, True)
    for p, (m, t, y) in enumerate(pts):
        ref = O.nlml_grad(kidx, Q, 1, 0, None, t, y, th[p])
        assert st[p] == 0
        assert_parity(nlml[p], grad[p], ref)
    ctx.close()


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "appendixA_*.npz"))))
def test_golden_fixtures(path):
    """Committed vectors: inputs of the survey's reference run; expected = oracle fp64 (tight).  (The fp32 scalars the survey
    session recorded from a shimmed-header build of the reference are in the same files; they are NOT evidence and are only
    looked at by test_recorded_fp32_reference_scalars_not_evidence below.)"""
    g = np.load(path)
    D, N, Q, R = int(g["D"]), int(g["N"]), int(g["Q"]), int(g["R"])
    ctx = make_ctx(7, Q, D, R, [(g["meta"], g["t"], g["y"])])
    nlml, grad, st = ctx.nlml_grad([0], g["theta"][None, :], True)
    assert st[0] == 0
    assert abs(nlml[0] - float(g["oracle_nlml"])) <= NLML_RTOL * abs(nlml[0])
    if "oracle_grad" in g:
        assert_parity(nlml[0], grad[0], {"nlml": float(g["oracle_nlml"]), "grad": g["oracle_grad"]})
    if "oracle_nlml_prior2" in g:
        ctx.set_prior(0, *synth.hier_gamma_prior(Q, D, R, 0.01))
        n2, g2, _ = ctx.nlml_grad([0], g["theta"][None, :], True)
        assert_parity(n2[0], g2[0], {"nlml": float(g["oracle_nlml_prior2"]), "grad": g["oracle_grad_prior2"]})
    ctx.close()


def test_recorded_fp32_reference_scalars_not_evidence():
    """RECORDED, NOT EVIDENCE.  Five nlml scalars printed by the reference in the survey session (a build that needed a stand-in
    mkl.h, which this project may not reproduce) sit in the appendixA fixtures.  They pin nothing -- parity rests on the
    reference-derived Gram fixtures, the oracle and the finite-difference tests -- and this test only keeps the record honest:
    the device values stay within the reference's own fp32 noise floor (1e-6) of them."""
    for path in sorted(glob.glob(os.path.join(GOLD, "appendixA_*.npz"))):
        g = np.load(path)
        D, N, Q, R = int(g["D"]), int(g["N"]), int(g["Q"]), int(g["R"])
        ctx = make_ctx(7, Q, D, R, [(g["meta"], g["t"], g["y"])])
        nlml, _, st = ctx.nlml_grad([0], g["theta"][None, :], False)
        assert st[0] == 0 and abs(nlml[0] - float(g["ref_fp32_nlml"])) <= 1e-6 * abs(nlml[0])
        if "ref_fp32_nlml_prior2" in g:
            ctx.set_prior(0, *synth.hier_gamma_prior(Q, D, R, 0.01))
            n2, _, _ = ctx.nlml_grad([0], g["theta"][None, :], False)
            assert abs(n2[0] - float(g["ref_fp32_nlml_prior2"])) <= 1e-6 * abs(n2[0])
        ctx.close()


def test_failure_semantics():
    D, Q, R = 2, 2, 2
    th = synth.theta(1, 0, 7, Q, D, R)
    tiny = (np.array([0, 1], np.int32), np.array([1, 2], np.float32), np.array([0, 1], np.float32))
    sing = (np.zeros(6, np.int32), np.array([1, 1, 1, 2, 2, 2], np.float32), np.ones(6, np.float32))
    good = synth.patient(1, 0, D, 40)
    ctx = make_ctx(7, Q, D, R, [tiny, sing, good])
    th2 = th.copy()
    th2[:D] = -80.0
    nlml, grad, st = ctx.nlml_grad([0, 1, 2], np.stack([th, th2, th]), True)
    assert st[0] == -1 and np.isnan(nlml[0]) and np.all(np.isnan(grad[0]))     # n <= 2 (c_objective_one.cpp:51)
    assert st[1] == -1 and np.isnan(nlml[1])                                     # 10 jitters exhausted
    ref = O.nlml_grad(7, Q, D, R, *good, th)
    assert st[2] == 0
    assert_parity(nlml[2], grad[2], ref)
    # argument errors come back as codes + message, never exit()
    with pytest.raises(medgp_amd.MedgpError):
        ctx.nlml_grad([7], th[None, :], True)
    with pytest.raises(medgp_amd.MedgpError):
        ctx.set_patient(0, np.array([0, 5, 1], np.int32), np.zeros(3, np.float32), np.zeros(3, np.float32))
    ctx.close()


def test_get_factor_and_predict():
    D, N, Q, R = 3, 90, 3, 2
    m, t, y = synth.patient(8, 0, D, N)
    th = synth.theta(8, 0, 7, Q, D, R)
    ctx = make_ctx(7, Q, D, R, [(m, t, y)])
    ctx.nlml_grad([0], th[None, :], True)
    alpha, linv, beta = ctx.get_factor(0, N)
    ref = O.nlml_grad(7, Q, D, R, m, t, y, th, want_alpha=True, want_linv=True)
    np.testing.assert_allclose(alpha, ref["alpha"], rtol=2e-6, atol=1e-6 * np.abs(ref["alpha"]).max())
    np.testing.assert_allclose(linv, ref["linv"], rtol=2e-6, atol=1e-6 * np.abs(ref["linv"]).max())
    assert abs(beta - ref["beta"]) <= 1e-6 * abs(ref["beta"])
    assert np.all(np.triu(linv, 1) == 0)
    m2 = np.array([0, 2, 1, 1], np.int32)
    t2 = np.array([3.5, 77.0, 150.25, float(t[5])], np.float32)
    mean, var, st = ctx.fit_predict(0, th, m2, t2)
    rp = O.fit_predict(7, Q, D, R, m, t, y, th, m2, t2)
    assert st == 0
    np.testing.assert_allclose(mean, rp["mean"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(var, rp["var"], rtol=1e-5, atol=1e-6)
    ctx.close()


def test_full_size_properties():
    """BASELINE sizes (D=24, N=512): properties that need no oracle run.
    (a) batch-composition invariance: a patient evaluated alone == inside a batch, bit for bit;
    (b) duplicate patients in different slots give identical bits; run-to-run reproducible;
    (c) observation-permutation invariance (<= 1e-10);
    (d) gradient vs central finite differences of the GPU's own nlml on a few hypers."""
    D, N, Q, R, P = 24, 512, 5, 8, 24
    pts, th = synth.cohort(31, P, D, N, Q=Q, R=R)
    pts[5] = pts[3]
    th[5] = th[3]
    ctx = make_ctx(7, Q, D, R, pts)
    nlml, grad, st = ctx.nlml_grad(np.arange(P), th, True)
    assert np.all(st == 0)
    n1, g1, _ = ctx.nlml_grad([3], th[3:4], True)
    assert n1[0] == nlml[3] and np.array_equal(g1[0], grad[3])
    assert nlml[5] == nlml[3] and np.array_equal(grad[5], grad[3])
    nlml_b, grad_b, _ = ctx.nlml_grad(np.arange(P), th, True)
    assert np.array_equal(nlml_b, nlml) and np.array_equal(grad_b, grad)
    # (c)
    m, t, y = pts[0]
    perm = np.random.default_rng(0).permutation(N)
    ctx.set_patient(1, m[perm], t[perm], y[perm])
    n2, g2, _ = ctx.nlml_grad([1], th[0:1], True)
    assert abs(n2[0] - nlml[0]) <= 1e-10 * abs(nlml[0])
    np.testing.assert_allclose(g2[0], grad[0], rtol=1e-7, atol=1e-9 * np.abs(grad[0]).max())
    # (d)
    H = th.shape[1]
    hs = [0, D - 1, D + 5, D + Q * D * R - 1, D + Q * D * R, D + Q * D * R + Q + 1, H - 1]
    eps = 1e-6
    tp = np.repeat(th[0:1], 2 * len(hs), axis=0)
    for k, h in enumerate(hs):
        tp[2 * k, h] += eps
        tp[2 * k + 1, h] -= eps
    nf, _, _ = ctx.nlml_grad(np.zeros(2 * len(hs), np.int32), tp, False)
    gs = np.abs(grad[0]).max()
    for k, h in enumerate(hs):
        fd = (nf[2 * k] - nf[2 * k + 1]) / (2 * eps)
        assert abs(fd - grad[0][h]) <= 5e-6 * max(abs(grad[0][h]), 1e-2 * gs), (h, fd, grad[0][h])
    ctx.close()


def test_device_gradient_all_hypers_richardson_fd():
    """Rows a14-a19 at the headline shape: EVERY one of the 1114 gradient components of one D=24, N=512 patient (the
    appendix-A fixture) against Richardson-extrapolated central differences (steps 2e-3 and 1e-3) of the device's own nlml --
    4 x 1114 nlml-only evaluations, batched 1114 per call.  The gradient kernels (k_wgrad / k_slabsum / k_epilogue) share
    nothing with the nlml-only route but the factorisation, so this pins them to the objective itself.
    Tolerance: 1e-6 relative to max(|g_h|, 1e-3 max|g|) (observed on the oracle side: 5e-8)."""
    g = np.load(os.path.join(GOLD, "appendixA_D24_N512.npz"))
    D, N, Q, R = int(g["D"]), int(g["N"]), int(g["Q"]), int(g["R"])
    th = g["theta"]
    H = th.size
    ctx = medgp_amd.Context(7, Q, D, R)
    ctx.reserve(1, N, H)
    ctx.set_patient(0, g["meta"], g["t"], g["y"])
    nl, grad, st = ctx.nlml_grad([0], th[None, :], True)
    assert st[0] == 0
    s = 2e-3
    slots = np.zeros(H, np.int32)
    vals = []
    for step in (s, -s, s / 2, -s / 2):
        tp = np.repeat(th[None, :], H, axis=0)
        tp[np.arange(H), np.arange(H)] += step
        nf, _, stf = ctx.nlml_grad(slots, tp, False)
        assert np.all(stf == 0)
        vals.append(nf)
    a = (vals[0] - vals[1]) / (2 * s)
    b = (vals[2] - vals[3]) / s
    fd = (4 * b - a) / 3
    gs = np.abs(grad[0]).max()
    err = np.abs(fd - grad[0]) / np.maximum(np.abs(grad[0]), 1e-3 * gs)
    assert err.max() <= 1e-6, (int(err.argmax()), err.max())
    # and the same components against the committed oracle gradient
    eo = np.abs(grad[0] - g["oracle_grad"]) / np.maximum(np.abs(g["oracle_grad"]), 1e-3 * gs)
    assert eo.max() <= GRAD_RTOL
    ctx.close()


def test_device_pointer_api_matches_host_api():
    import torch
    D, N, Q, R, P = 2, 256, 5, 2, 16
    pts, th = synth.cohort(5, P, D, N, Q=Q, R=R)
    ctx = make_ctx(7, Q, D, R, pts)
    nlml, grad, st = ctx.nlml_grad(np.arange(P), th, True)
    dev = torch.device("cuda", 0)
    th_d = torch.from_numpy(th).to(dev)
    nl_d = torch.empty(P, dtype=torch.float64, device=dev)
    g_d = torch.empty((P, ctx.H), dtype=torch.float64, device=dev)
    s_d = torch.empty(P, dtype=torch.int32, device=dev)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.nlml_grad_device(np.arange(P), th_d.data_ptr(), True, nl_d.data_ptr(), g_d.data_ptr(), s_d.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(nl_d.cpu().numpy(), nlml) and np.array_equal(g_d.cpu().numpy(), grad)
    assert np.array_equal(s_d.cpu().numpy(), st)
    ctx.set_stream(None)
    ctx.close()


def test_baseline_config2_full_size():
    """BASELINE config 2: 256 synthetic patients x N=256, D=2, fp64, one batch on one MI355X."""
    D, N, Q, R, P = 2, 256, 5, 2, 256
    pts, th = synth.cohort(202, P, D, N, Q=Q, R=R)
    ctx = make_ctx(7, Q, D, R, pts)
    nlml, grad, st = ctx.nlml_grad(np.arange(P), th, True)
    assert np.all(st == 0) and np.all(np.isfinite(nlml)) and np.all(np.isfinite(grad))
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:     # ALL 256 patients against the oracle
        refs = list(ex.map(lambda p: O.nlml_grad(7, Q, D, R, *pts[p], th[p]), range(P)))
    for p, ref in enumerate(refs):
        assert ref["status"] == 0
        assert_parity(nlml[p], grad[p], ref, f"p{p}")
    ctx.close()


def test_config5_shape_and_many_components():
    """D = 64 outputs (BASELINE config 5's width; B_q table 164 KB) at a reduced N; Q = 9, 12 and 16 (the tuned pair kernels in two
    launches: the first eight components, then the rest into the same tiles / their own slab planes) and Q = 17 > 16, which takes
    the generic (non-templated) assembly / gradient kernels.  Q is a free configuration key of the reference
    (ref: kernel/c_kernel_LMC_SM.cpp:51-70)."""
    for (D, N, Q, R) in ((64, 320, 5, 8), (3, 150, 9, 2), (3, 200, 12, 2), (2, 130, 16, 2), (3, 150, 17, 2)):
        pts, th = synth.cohort(303, 2, D, N, Q=Q, R=R)
        ctx = make_ctx(7, Q, D, R, pts)
        nlml, grad, st = ctx.nlml_grad([0, 1], th, True)
        for p in range(2):
            ref = O.nlml_grad(7, Q, D, R, *pts[p], th[p], nthreads=4)
            assert st[p] == 0
            assert_parity(nlml[p], grad[p], ref, f"D{D} Q{Q} p{p}")
        ctx.close()


def test_multi_cu_and_single_workgroup_factorisations_agree(monkeypatch):
    """The multi-CU panel path (few large patients) and the one-workgroup-per-patient path are two schedules of the
    same arithmetic: results must agree to rounding, and each must match the oracle."""
    D, N, Q, R = 24, 700, 5, 8
    pts, th = synth.cohort(404, 2, D, N, Q=Q, R=R)
    out = {}
    for mode in ("1", "-1"):
        monkeypatch.setenv("MEDGP_MULTI_CU", mode)
        ctx = make_ctx(7, Q, D, R, pts)
        out[mode] = ctx.nlml_grad([0, 1], th, True)
        ctx.close()
    ref = O.nlml_grad(7, Q, D, R, *pts[0], th[0], nthreads=8)
    for mode in out:
        assert_parity(out[mode][0][0], out[mode][1][0], ref, f"mode{mode}")
    np.testing.assert_allclose(out["1"][0], out["-1"][0], rtol=1e-12)


def test_fit_predict_batch_matches_single_calls():
    D, Q, R = 3, 3, 2
    pts = [synth.patient(71, p, D, n) for p, n in enumerate((1, 2, 30, 77))]   # n = 1, 2: no n > 2 guard on this path
    th = np.stack([synth.theta(71, p, 7, Q, D, R) for p in range(4)])
    ctx = make_ctx(7, Q, D, R, pts)
    m2 = np.array([0, 1, 2, 1], np.int32)
    t2 = np.array([3.0, 50.0, 120.5, 77.25], np.float32)
    mean, var, st = ctx.fit_predict_batch([0, 1, 2, 3], th, m2, t2)
    assert np.all(st == 0)
    for p in range(4):
        rp = O.fit_predict(7, Q, D, R, *pts[p], th[p], m2[p:p + 1], t2[p:p + 1])
        np.testing.assert_allclose(mean[p], rp["mean"][0], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(var[p], rp["var"][0], rtol=1e-5, atol=1e-6)
        m1, v1, s1 = ctx.fit_predict(p, th[p], m2[p:p + 1], t2[p:p + 1])
        assert m1[0] == mean[p] and v1[0] == var[p]
    ctx.close()


@pytest.mark.parametrize("P,N,force_single", [
    (67, 130, True),     # >= 64 entries, not a multiple of 8: XCD-local tile order of k_wgrad with a ragged last group
    (261, 200, False),   # more entries than CUs: two 4-wave workgroups per CU, several passes per step
    (5, 330, True),      # few entries on the single-workgroup kernel: 8-wave shape, spread tile order
])
def test_batch_shapes_of_the_kernel_variants(P, N, force_single, monkeypatch):
    """Every launch-shape branch of k_cholinv / k_wgrad against the oracle: ragged n (each patient has its own n <= N),
    batch sizes on both sides of the 64-entry and #CU thresholds."""
    if force_single:
        monkeypatch.setenv("MEDGP_MULTI_CU", "-1")
    D, Q, R = 6, 3, 2
    rng = np.random.default_rng(5)
    pts, th = synth.cohort(303, P, D, N, Q=Q, R=R)
    pts = [(m[:k], t[:k], y[:k]) for (m, t, y), k in zip(pts, rng.integers(N // 2, N + 1, size=P))]
    ctx = make_ctx(7, Q, D, R, pts)
    nlml, grad, st = ctx.nlml_grad(np.arange(P), th, True)
    nlml0, _, st0 = ctx.nlml_grad(np.arange(P), th, False)
    assert (st == 0).all() and (st0 == 0).all()
    assert np.array_equal(nlml0, nlml)
    for p in list(range(0, P, max(P // 6, 1))) + [P - 1]:
        m, t, y = pts[p]
        assert_parity(nlml[p], grad[p], O.nlml_grad(7, Q, D, R, m, t, y, th[p], nthreads=4), f"p{p}")
    ctx.close()


@pytest.mark.parametrize("case", ["PT_INR_mode2", "PT_INR_mode2_varem", "PT_INR_mode2_testclamp", "PT_INR_mode0_testclamp",
                                  "all24_mode2_varem", "all24_mode2_testclamp", "D64_mode2_varem"])
def test_device_prior_stage_vs_reference_compiled_prior(case):
    """k_epilogue's prior stage (row a19) against numbers the REFERENCE's own compiled code produced (round 6):
    inference/c_inference_prior.cpp + prior/c_prior.cpp build unmodified with plain g++; oracle/ref_prior_inference_dump.cpp ran
    c_inference_prior::compute_nlml on prior objects built by the reference's setup_param / init_test_prior / the variational-EM
    writes and recorded, per hyper, what it did to a base (nlml, gradient): tests/golden/ref_prior_inference.json.gz.  The prior stage is
    additive, so on the device (with prior) - (without prior) at the fixture's theta must be the reference's shift: nlml by - sum lp,
    gradients by - hyp dlp (exp chain rule) or - dlp, clamped entries exactly 0 -- including the reference's SINGLE-precision
    log(2 b) in the Laplace normaliser (ref: prior/c_prior.cpp:404), which only this comparison could find."""
    import gzip
    import json
    with gzip.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_prior_inference.json.gz"), "rt") as f:
        c = next(x for x in json.load(f)["cases"] if x["name"] == case)
    Q, D, R = c["Q"], c["D"], c["R"]
    H = len(c["theta"])
    th = np.array(c["theta"])[None, :]
    flag = np.array(c["prior_flag"], np.uint8)
    typ = np.array(c["prior_type"], np.int32)
    ex = np.array(c["prior_exp"], np.uint8)
    p0 = np.array(c["prior_p0"], np.float32)
    p1 = np.array(c["prior_p1"], np.float32)
    N = 3 * D + 40
    m, t, y = synth.patient(77, 0, D, N)
    ctx = medgp_amd.Context(7, Q, D, R)
    assert ctx.H == H
    ctx.reserve(1, N, 1)
    ctx.set_patient(0, m, t, y)
    n0, g0, s0 = ctx.nlml_grad([0], th, True)
    ctx.set_prior(0, flag, typ, ex, p0, p1)
    n1, g1, s1 = ctx.nlml_grad([0], th, True)
    n1b, _, _ = ctx.nlml_grad([0], th, False)
    ctx.close()
    assert s0[0] == 0 and s1[0] == 0
    ref_dn = c["nlml"] - c["base_nlml"]                      # = - sum of the reference's lp
    ref_dg = np.array(c["dnlml"]) - np.array(c["base_dnlml"])
    assert abs((n1[0] - n0[0]) - ref_dn) <= 1e-10 * max(abs(n1[0]), abs(ref_dn), 1.0), (n1[0] - n0[0], ref_dn)
    assert abs(n1b[0] - n1[0]) <= 1e-12 * abs(n1[0])
    clamp = (flag == 1) & (typ == 0)
    assert (g1[0][clamp] == 0.0).all() and (np.array(c["dnlml"])[clamp] == 0.0).all()
    free = ~clamp
    err = np.abs((g1[0] - g0[0])[free] - ref_dg[free])
    assert np.all(err <= 1e-10 * np.maximum(1.0, np.abs(g1[0][free]))), float(err.max())
    if "mode2" in case:
        assert np.any(ref_dg[free] != 0.0) and ref_dn != 0.0
    else:
        assert ref_dn == 0.0 and not np.any(ref_dg[free] != 0.0)     # mode 0: only the test-time clamp acts
