"""CPU restatement of the cohort mode estimation (TEST INFRASTRUCTURE ONLY, like everything under oracle/).

    compute_kde / compute_mode   <- medgpc/clustering/mode_estimate.py:438-450
    output_mode_lmc_sm           <- medgpc/clustering/mode_estimate.py:242-435 (numerical part: no directories, files, plots)
    output_mode_se / _sm         <- medgpc/clustering/mode_estimate.py:29-79 / :82-240

The reference's KDE is statsmodels' KDEUnivariate(kernel="gau", bw="silverman"); statsmodels is a third-party dependency
(README.md:10, no version pinned) that is absent from /root/reference and from this image, so its PUBLISHED algorithm is
restated here (statsmodels/nonparametric: bandwidths.bw_silverman / _select_sigma, kernels.Gaussian, CustomKernel.density):
    A = min(std(x, ddof=1), IQR / 1.349) if IQR > 0 else std(x, ddof=1),  IQR from scipy.stats.scoreatpercentile(x, 75 / 25)
    h = 0.9 A n^(-1/5);   evaluate(p) = 1/h * mean_j phi((x_j - p) / h),  phi(u) = 0.3989422804014327 exp(-u^2 / 2)
Pins (tests/test_cohort_mode.py): the density against scipy.stats.gaussian_kde driven at the same bandwidth (an independent
implementation of the same estimator), the percentiles against scipy.stats.scoreatpercentile.  The bandwidth RULE itself can
only be pinned to the published formula: parity of the rule is UNPINNED against a statsmodels run.
"""
import numpy as np


def scoreatpercentile(x, per):
    """scipy.stats.scoreatpercentile(x, per) with the default 'fraction' interpolation."""
    v = np.sort(np.asarray(x, dtype=np.float64))
    idx = per / 100.0 * (len(v) - 1)
    lo = int(np.floor(idx))
    hi = min(lo + 1, len(v) - 1)
    return v[lo] + (idx - lo) * (v[hi] - v[lo])


def silverman_bw(x):
    """statsmodels bw_silverman (gaussian kernel): 0.9 A n^-0.2 with _select_sigma's A."""
    x = np.asarray(x, dtype=np.float64)
    iqr = (scoreatpercentile(x, 75) - scoreatpercentile(x, 25)) / 1.349
    sd = np.std(x, ddof=1)
    A = min(sd, iqr) if iqr > 0 else sd
    return 0.9 * A * len(x) ** (-0.2)


def kde_density(data, test_x, h):
    """CustomKernel.density for the Gaussian kernel: 1/h * mean_j phi((x_j - p)/h) at every p in test_x."""
    data = np.asarray(data, dtype=np.float64).ravel()
    test_x = np.asarray(test_x, dtype=np.float64).ravel()
    out = np.empty(len(test_x))
    for k in range(0, len(test_x), 512):          # blocked so that a 4096 x 4096 table is never held
        u = (data[:, None] - test_x[None, k:k + 512]) / h
        out[k:k + 512] = np.mean(0.3989422804014327 * np.exp(-u * u / 2.0), axis=0) / h
    return out


def compute_kde(data, test_x):
    """ref: mode_estimate.py:438-444.  Raises like KDEUnivariate.fit when the bandwidth is not positive."""
    data = np.asarray(data, dtype=np.float64).ravel()
    if len(data) < 2 or not np.all(np.isfinite(data)):
        raise RuntimeError("KDE needs at least two finite samples")
    h = silverman_bw(data)
    if not h > 0:
        raise RuntimeError("Selected KDE bandwidth is 0. Cannot estimate density.")
    return kde_density(data, test_x, h), h


def compute_mode(data, density, weighted=True):
    """ref: mode_estimate.py:446-450."""
    data = np.asarray(data, dtype=np.float64).ravel()
    if weighted:
        return np.nansum(data * density) / np.nansum(density)
    return data[np.argmax(density)]


def kde_mode(data, weighted=True, test=None):
    pts = data if test is None else test
    dens, _ = compute_kde(data, pts)
    return compute_mode(pts, dens, weighted)


def output_mode_se(hyp_array, mode_fn=kde_mode):
    """ref: mode_estimate.py:46-61 (hypers: nugget, lengthscale, scalefactor)."""
    hyp_array = np.asarray(hyp_array, dtype=np.float64)
    out = np.zeros(hyp_array.shape[1])
    for i in range(hyp_array.shape[1]):
        x = np.exp(hyp_array[:, i])
        if i == 1:                                                   # lengthscale: on a grid, ref :54-57
            out[i] = np.log(mode_fn(x, False, np.linspace(0.01, 1000.0, 100001)))
        else:
            out[i] = np.log(mode_fn(x, False))
    return out


def output_mode_sm(Q, pan_array, hyp_array, mixture_pan, mixture_index, mixture_cluster_num, mixture_cluster_assign, mode_fn=kde_mode):
    """ref: mode_estimate.py:100-226 (D = 1; hypers: nugget, w_q, mu_q, sqrt v_q)."""
    pan_array = np.asarray(pan_array)
    hyp_array = np.asarray(hyp_array, dtype=np.float64)
    mixture_pan, mixture_index = np.asarray(mixture_pan), np.asarray(mixture_index)
    mixture_cluster_assign = np.asarray(mixture_cluster_assign)
    newQ = int(mixture_cluster_num)
    out = np.zeros(1 + 3 * newQ)
    out[0] = np.log(mode_fn(np.exp(hyp_array[:, 0]), False))        # ref :108-113
    row_of = {p: i for i, p in enumerate(pan_array.tolist())}
    cluster_ids = np.unique(mixture_cluster_assign)
    assert len(cluster_ids) == newQ
    per = np.linspace(0.01, 1000.0, 100001)
    for q, cid in enumerate(cluster_ids):
        comp = np.where(mixture_cluster_assign == cid)[0]
        all_mu = np.array([np.exp(hyp_array[row_of[mixture_pan[c]], 1 + Q + mixture_index[c]]) for c in comp])
        all_v = np.array([np.exp(hyp_array[row_of[mixture_pan[c]], 1 + 2 * Q + mixture_index[c]]) for c in comp])
        out[1 + newQ + q] = np.log(mode_fn(all_mu, False, 1.0 / per))                       # ref :170-174
        out[1 + 2 * newQ + q] = np.log(mode_fn(all_v, False, 1.0 / (2.0 * np.pi * per)))    # ref :181-185
        cpan, cidx = mixture_pan[comp], mixture_index[comp]
        all_w = np.array([sum(np.exp(hyp_array[row_of[pan], 1 + qq]) for qq in cidx[cpan == pan]) for pan in np.unique(cpan)])
        out[1 + q] = np.log(mode_fn(all_w, False))                                           # ref :198-226
    return out


def output_mode_lmc_sm(Q, D, R, pan_array, hyp_array, mixture_pan, mixture_index, mixture_cluster_num, mixture_cluster_assign,
                       mode_fn=kde_mode):
    """ref: mode_estimate.py:262-424, the numbers only.  Returns kde_mode_hyp for newQ = mixture_cluster_num components."""
    pan_array = np.asarray(pan_array)
    hyp_array = np.asarray(hyp_array, dtype=np.float64)
    mixture_pan = np.asarray(mixture_pan)
    mixture_index = np.asarray(mixture_index)
    mixture_cluster_assign = np.asarray(mixture_cluster_assign)
    newQ = int(mixture_cluster_num)
    out = np.zeros(D + newQ * (D * R + 2 + D))
    for d in range(D):                                               # ref :273-279
        out[d] = np.log(mode_fn(np.exp(hyp_array[:, d])))
    cluster_ids = np.unique(mixture_cluster_assign)                  # ref :287-289
    assert len(cluster_ids) == newQ
    row_of = {p: i for i, p in enumerate(pan_array.tolist())}
    for q, cid in enumerate(cluster_ids):                            # ref :318
        comp = np.where(mixture_cluster_assign == cid)[0]
        all_mu = np.array([np.exp(hyp_array[row_of[mixture_pan[c]], D + Q * D * R + mixture_index[c]]) for c in comp])
        all_v = np.array([np.exp(hyp_array[row_of[mixture_pan[c]], D + Q * D * R + Q + mixture_index[c]]) for c in comp])
        out[D + newQ * D * R + q] = np.log(mode_fn(all_mu))          # ref :339-342
        out[D + newQ * (D * R + 1) + q] = np.log(mode_fn(all_v))     # ref :350-353
        cpan, cidx = mixture_pan[comp], mixture_index[comp]
        all_B = []
        for pan in np.unique(cpan):                                  # ref :369-386
            hyp = hyp_array[row_of[pan]]
            B = np.zeros((D, D))
            for qq in cidx[cpan == pan]:
                A = hyp[D + qq * D * R: D + (qq + 1) * D * R].reshape(D, R)
                lam = np.exp(hyp[D + Q * (D * R + 2) + qq * D: D + Q * (D * R + 2) + (qq + 1) * D])
                B += A @ A.T + np.diag(lam)
            all_B.append(B)
        all_B = np.asarray(all_B)
        kde_B = np.zeros((D, D))
        for d1 in range(D):                                          # ref :406-413
            for d2 in range(d1, D):
                kde_B[d1, d2] = kde_B[d2, d1] = mode_fn(all_B[:, d1, d2])
        U, S, _ = np.linalg.svd(kde_B)                               # ref :423-431
        A_ = (U * np.sqrt(S))[:, 0:R]
        lam_ = np.diag(kde_B - A_ @ A_.T).copy()
        lam_[lam_ <= 0.0] = 1e-15
        out[D + newQ * (D * R + 2) + q * D: D + newQ * (D * R + 2) + (q + 1) * D] = np.log(lam_)
        out[D + q * D * R: D + (q + 1) * D * R] = A_.reshape(-1)
    return out
