"""medgp_kde_mode (HIP) against the oracle, through the C ABI (ref: mode_estimate.py:438-450).
Tolerance: 1e-12 relative on bandwidth and mode (fp64 sums in a different order than numpy's; observed ~1e-15)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
pytestmark = pytest.mark.gpu
RTOL = 1e-12


def _oracle(series, weighted):
    from oracle import kde_oracle as KO
    return np.array([KO.kde_mode(s, weighted) for s in series]), np.array([KO.silverman_bw(s) for s in series])


def test_kde_mode_ragged_series_vs_oracle():
    from medgp_amd import capi
    rng = np.random.default_rng(11)
    series = [rng.normal(size=n) * s + m for n, s, m in
              ((2, 1, 0), (3, 2, 1), (5, 1e-3, 7), (64, 1, -3), (257, 10, 100), (1000, 0.5, 0), (2049, 1, 0), (4096, 3, 2))]
    series.append(np.exp(rng.normal(size=700)))                       # skewed, like exp(log-hypers)
    series.append(np.round(rng.normal(size=300) * 2) / 2)             # heavy ties (ranks decided by index)
    series.append(np.array([1.0] * 7 + [5.0]))                        # IQR = 0: bandwidth from the standard deviation
    series.append(np.concatenate([rng.normal(size=50), [1e6]]))       # an outlier whose Gaussian terms underflow
    for weighted in (True, False):
        mode, bw, st, ms = capi.kde_mode(series, weighted, full=True)
        want, wbw = _oracle(series, weighted)
        assert np.all(st == 0) and ms > 0
        np.testing.assert_allclose(bw, wbw, rtol=RTOL)
        if weighted:
            np.testing.assert_allclose(mode, want, rtol=RTOL, atol=1e-12 * np.abs(want).max())
        else:
            assert np.array_equal(mode, want)                          # the arg-max mode is one of the samples


def test_kde_mode_on_grids_vs_oracle():
    """medgp_kde_mode_at: densities on per-series grids (the reference's 100001-point length-scale / period grids), mixed with
    series evaluated at their samples."""
    from medgp_amd import capi
    from oracle import kde_oracle as KO
    rng = np.random.default_rng(17)
    per = np.linspace(0.01, 1000.0, 100001)
    series = [1.0 / rng.uniform(12, 72, 300), rng.normal(size=100), 1.0 / (2 * np.pi * rng.uniform(6, 72, 77)), rng.uniform(5, 80, 40)]
    grids = [1.0 / per, None, 1.0 / (2 * np.pi * per), per]
    for weighted in (False, True):
        mode, bw, st, _ = capi.kde_mode(series, weighted, full=True, test=grids)
        assert np.all(st == 0)
        want = np.array([KO.kde_mode(s, weighted, g) for s, g in zip(series, grids)])
        if weighted:
            np.testing.assert_allclose(mode, want, rtol=RTOL)
        else:
            # arg-max over a fine grid: the densities of neighbouring points differ by less than rounding near the top, so
            # compare the density AT the chosen points instead of the indices
            for s, g, m, w in zip(series, grids, mode, want):
                pts = s if g is None else g
                assert m in pts
                h = KO.silverman_bw(s)
                dm, dw = KO.kde_density(s, np.array([m]), h)[0], KO.kde_density(s, np.array([w]), h)[0]
                assert abs(dm - dw) <= 1e-13 * dw


def test_output_mode_se_sm_on_device(tmp_path):
    from medgp_amd import cohort_mode
    from oracle import kde_oracle as KO
    from test_cohort_mode import make_sm_cohort
    c = make_sm_cohort(8, P=60, Q=3, newQ=2)
    exp = dict(c["exp"], exp_kernel_dir=str(tmp_path / "sm"))
    got = cohort_mode.output_mode_kernel(-1, exp, c["pan"], c["hyp"], c["mpan"], c["midx"], 2, c["assign"], "kmeans")
    want = KO.output_mode_sm(3, c["pan"], c["hyp"], c["mpan"], c["midx"], 2, c["assign"])
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-4)    # grid spacing 0.01 h: a tie at the top may move one grid point
    assert np.array_equal(got[:3], want[:3])                     # nugget and weights: arg max over the samples, exact


def test_kde_mode_failures_mirror_the_reference_fit():
    from medgp_amd import capi, cohort_mode
    series = [np.ones(9), np.array([3.0]), np.array([1.0, np.nan, 2.0]), np.array([1.0, np.inf]), np.arange(6.0)]
    mode, bw, st, _ = capi.kde_mode(series, True, full=True)
    assert st.tolist() == [-1, -1, -1, -1, 0] and np.all(np.isnan(mode[:4])) and np.isfinite(mode[4])
    with pytest.raises(capi.MedgpError):
        cohort_mode.kde_modes(series)
    assert capi.kde_mode([]).shape == (0,)


def test_kde_mode_cohort_size_properties():
    """P = 4096 subjects, D = 24: the 300 element series of one cluster.  The oracle needs minutes there, so: equivariance
    under x -> a x + b (bandwidth scales with |a|), invariance under permutation of the samples, and agreement with the
    oracle on a few of the series."""
    from medgp_amd import capi
    rng = np.random.default_rng(5)
    P, ns = 4096, 300
    base = [rng.normal(size=P) * rng.uniform(0.1, 3) + rng.uniform(-2, 2) for _ in range(ns)]
    m0, b0, st, ms = capi.kde_mode(base, True, full=True)
    assert np.all(st == 0)
    a, b = -2.5, 0.75
    m1, b1, _, _ = capi.kde_mode([a * x + b for x in base], True, full=True)
    np.testing.assert_allclose(m1, a * m0 + b, rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(b1, abs(a) * b0, rtol=1e-12)
    m2 = capi.kde_mode([x[rng.permutation(P)] for x in base], True)
    np.testing.assert_allclose(m2, m0, rtol=1e-12, atol=1e-13)
    want, wbw = _oracle(base[:3], True)
    np.testing.assert_allclose(m0[:3], want, rtol=RTOL, atol=1e-13)
    np.testing.assert_allclose(b0[:3], wbw, rtol=RTOL)
    # repeatable bit for bit (fixed-order sums)
    assert np.array_equal(capi.kde_mode(base[:20], True), m0[:20])


def test_output_mode_lmc_sm_on_device_vs_oracle(tmp_path):
    from medgp_amd import cohort_mode
    from oracle import kde_oracle as KO
    from test_cohort_mode import make_cohort
    c = make_cohort(9, P=120, Q=3, D=5, R=2, newQ=3)
    exp = dict(c["exp"], exp_kernel_dir=str(tmp_path / "kern"))
    got = cohort_mode.output_mode_kernel(-1, exp, c["pan"], c["hyp"], c["mpan"], c["midx"], 3, c["assign"], "kmeans")
    want = KO.output_mode_lmc_sm(3, 5, 2, c["pan"], c["hyp"], c["mpan"], c["midx"], 3, c["assign"])
    # modes agree to 1e-12; the SVD factors A_ inherit the conditioning of kde_B (columns are fixed up to that)
    np.testing.assert_allclose(got, want, rtol=1e-8, atol=1e-9)
    D, R, nq = 5, 2, 3
    np.testing.assert_allclose(got[:D], want[:D], rtol=1e-12)
    np.testing.assert_allclose(got[D + nq * D * R:D + nq * (D * R + 2)], want[D + nq * D * R:D + nq * (D * R + 2)], rtol=1e-12)
    assert np.array_equal(np.fromfile(tmp_path / "kern" / "all" / "kmeans_mode_param.bin"), got)
