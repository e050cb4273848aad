"""Cohort mode estimation, CPU side: the oracle against independent implementations, the host
logic of medgp_amd.cohort_mode against the oracle's restatement of output_mode_LMC_SM, the two-rank (gloo) path.
ref: medgpc/clustering/mode_estimate.py:242-450."""
import os
import subprocess
import sys

import numpy as np
import pytest
import scipy.stats

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from medgp_amd import cohort_mode  # noqa: E402
from oracle import kde_oracle as KO  # noqa: E402


def oracle_fn(series, weighted, tests=None):
    """The oracle as a stand-in for the HIP kernels (kde_fn signature of medgp_amd.cohort_mode)."""
    return np.array([KO.kde_mode(s, weighted, None if tests is None else tests[i]) for i, s in enumerate(series)])


def make_cohort(seed, P, Q, D, R, newQ):
    """Trained hypers of P subjects (string ids, like the reference's cohort id lists) and a component clustering in which
    some subjects have two components in one cluster and some none (both happen in the reference, :360-386)."""
    rng = np.random.default_rng(seed)
    H = D + Q * (D * R + 2 + D)
    hyp = np.empty((P, H))
    hyp[:, :D] = np.log(rng.uniform(0.15, 0.4, (P, D)))
    hyp[:, D:D + Q * D * R] = rng.uniform(-1.5, 1.5, (P, Q * D * R)) * 0.9 / np.sqrt(Q * R)
    hyp[:, D + Q * D * R:D + Q * D * R + Q] = np.log(1.0 / rng.uniform(12, 72, (P, Q)))
    hyp[:, D + Q * D * R + Q:D + Q * D * R + 2 * Q] = np.log(1.0 / (2 * np.pi * rng.uniform(6, 72, (P, Q))))
    hyp[:, D + Q * (D * R + 2):] = np.log(rng.uniform(0.1, 0.5, (P, Q * D)) * 0.1 / Q)
    pan = np.array([f"S{k:04d}" for k in rng.permutation(P)])
    mpan = np.repeat(pan, Q)
    midx = np.tile(np.arange(Q), P)
    assign = rng.integers(0, newQ, P * Q) * 3 + 1          # cluster ids need not be 0..newQ-1
    assign[:newQ] = np.arange(newQ) * 3 + 1                # every cluster used
    keep = rng.permutation(P * Q)                          # component order is arbitrary
    return dict(pan=pan, hyp=hyp, mpan=mpan[keep], midx=midx[keep], assign=assign[keep],
                exp={"kernel": "LMC-SM", "Q": Q, "D": D, "R": R})


def test_percentile_and_density_against_scipy():
    rng = np.random.default_rng(0)
    for n in (2, 3, 10, 257):
        x = rng.normal(size=n) * 3 + 1
        for per in (25, 75, 10):
            assert KO.scoreatpercentile(x, per) == pytest.approx(scipy.stats.scoreatpercentile(x, per), rel=1e-15, abs=1e-15)
        h = KO.silverman_bw(x)
        g = scipy.stats.gaussian_kde(x, bw_method=h / np.std(x, ddof=1))    # same estimator, driven at the same bandwidth
        pts = np.concatenate([x, rng.normal(size=7)])
        np.testing.assert_allclose(KO.kde_density(x, pts, h), g(pts), rtol=1e-12)
        grid = np.linspace(-5, 5, 41)                                      # grid evaluation: mode = the densest grid point
        assert KO.kde_mode(x, False, grid) == grid[np.argmax(g(grid))]


def test_silverman_rule_known_answers():
    # x = 0..4: std 1.58114, quartiles 1 and 3 -> IQR/1.349 = 1.48258 < std -> h = 0.9 * 1.48258 * 5^-0.2
    assert KO.silverman_bw(np.arange(5.0)) == pytest.approx(0.9 * (2 / 1.349) * 5 ** -0.2, rel=1e-15)
    # more than half the samples equal: IQR = 0 -> the standard deviation is used
    x = np.array([1.0] * 7 + [5.0])
    assert KO.silverman_bw(x) == pytest.approx(0.9 * np.std(x, ddof=1) * 8 ** -0.2, rel=1e-15)
    with pytest.raises(RuntimeError):
        KO.compute_kde(np.ones(5), np.ones(5))
    # the weighted "mode" of a symmetric sample is its centre; the arg-max mode is a sample
    x = np.array([-2.0, -1.0, 0.0, 1.0, 2.0])
    assert KO.kde_mode(x, True) == pytest.approx(0.0, abs=1e-15)
    assert KO.kde_mode(x, False) == 0.0


def test_deal_series_is_balanced_and_deterministic():
    costs = [n * n for n in (100, 5, 80, 80, 7, 60, 3, 90)]
    o = cohort_mode.deal_series(costs, 3)
    assert np.array_equal(o, cohort_mode.deal_series(costs, 3)) and set(o.tolist()) == {0, 1, 2}
    load = [sum(c for c, r in zip(costs, o) if r == k) for k in range(3)]
    assert max(load) <= 1.34 * sum(costs) / 3
    assert np.array_equal(cohort_mode.deal_series([4, 1], 1), [0, 0])


def test_output_mode_lmc_sm_host_logic_and_files(tmp_path):
    c = make_cohort(1, P=31, Q=3, D=4, R=2, newQ=2)
    calls = []

    def fn(series, weighted, tests=None):
        calls.append((len(series), weighted))
        return oracle_fn(series, weighted, tests)

    exp = dict(c["exp"], exp_kernel_dir=str(tmp_path / "kern"))
    got = cohort_mode.output_mode_kernel(2, exp, c["pan"], c["hyp"], c["mpan"], c["midx"], 2, c["assign"], "kmeans", kde_fn=fn)
    want = KO.output_mode_lmc_sm(3, 4, 2, c["pan"], c["hyp"], c["mpan"], c["midx"], 2, c["assign"])
    assert np.array_equal(got, want)
    assert calls == [(4 + 2 * (2 + 10), True)]             # ONE batched call: D nuggets + per cluster mu, v, D(D+1)/2 elements
    d = tmp_path / "kern" / "fold2"
    assert (d / "kmeans_mode_mixture_num.txt").read_text().split() == ["2"]
    assert np.array_equal(np.fromfile(d / "kmeans_mode_param.bin", dtype=np.float64), want)
    # layout of the result: the test-time reader expects D + newQ (D R + 2 + D) doubles (ref: c_experiment.cpp:179-219)
    assert len(want) == 4 + 2 * (4 * 2 + 2 + 4)
    with pytest.raises(NotImplementedError):
        cohort_mode.output_mode_kernel(-1, dict(exp, kernel="Matern"), c["pan"], c["hyp"], c["mpan"], c["midx"], 2, c["assign"], "kmeans", kde_fn=fn)


def make_sm_cohort(seed, P, Q, newQ):
    rng = np.random.default_rng(seed)
    hyp = np.empty((P, 1 + 3 * Q))
    hyp[:, 0] = np.log(rng.uniform(0.15, 0.4, P))
    hyp[:, 1:1 + Q] = np.log(rng.uniform(0.05, 1.0, (P, Q)))
    hyp[:, 1 + Q:1 + 2 * Q] = np.log(1.0 / rng.uniform(12, 72, (P, Q)))
    hyp[:, 1 + 2 * Q:] = np.log(1.0 / (2 * np.pi * rng.uniform(6, 72, (P, Q))))
    pan = np.array([f"S{k:04d}" for k in rng.permutation(P)])
    assign = rng.integers(0, newQ, P * Q)
    assign[:newQ] = np.arange(newQ)
    keep = rng.permutation(P * Q)
    return dict(pan=pan, hyp=hyp, mpan=np.repeat(pan, Q)[keep], midx=np.tile(np.arange(Q), P)[keep], assign=assign[keep],
                exp={"kernel": "SM", "Q": Q, "D": 1, "R": 1})


def test_output_mode_se_and_sm_host_logic(tmp_path):
    """The univariate families: arg-max modes, length-scale / period densities on the reference's 100001-point grids."""
    rng = np.random.default_rng(4)
    hyp = np.log(np.column_stack([rng.uniform(0.1, 0.5, 25), rng.uniform(5, 80, 25), rng.uniform(0.5, 2, 25)]))
    pan = np.arange(25)
    exp = {"kernel": "SE", "exp_kernel_dir": str(tmp_path / "se")}
    got = cohort_mode.output_mode_kernel(-1, exp, pan, hyp, pan, np.zeros(25, int), 1, np.zeros(25, int), "none", kde_fn=oracle_fn)
    assert np.array_equal(got, KO.output_mode_se(hyp))
    assert np.exp(got[1]) in np.linspace(0.01, 1000.0, 100001)          # the length-scale mode is a grid point
    assert np.array_equal(np.fromfile(tmp_path / "se" / "all" / "none_mode_param.bin"), got)
    c = make_sm_cohort(6, P=21, Q=3, newQ=2)
    exp = dict(c["exp"], exp_kernel_dir=str(tmp_path / "sm"))
    got = cohort_mode.output_mode_kernel(0, exp, c["pan"], c["hyp"], c["mpan"], c["midx"], 2, c["assign"], "kmeans", kde_fn=oracle_fn)
    want = KO.output_mode_sm(3, c["pan"], c["hyp"], c["mpan"], c["midx"], 2, c["assign"])
    assert np.array_equal(got, want) and len(got) == 1 + 3 * 2
    assert (tmp_path / "sm" / "fold0" / "kmeans_mode_mixture_num.txt").read_text().split() == ["2"]


def test_product_path_has_no_cpu_evaluator():
    """Without a GPU the default kde_fn must raise (library error), never compute on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from medgp_amd import capi
    with pytest.raises(capi.MedgpError):
        cohort_mode.kde_modes([np.arange(5.0)])


def test_cohort_mode_two_ranks_gloo(tmp_path):
    env = dict(os.environ, PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29553",
                          os.path.join(ROOT, "tests", "gloo_cohort_mode_worker.py"), str(tmp_path / "k")],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "GLOO_COHORT_MODE_OK" in out.stdout
