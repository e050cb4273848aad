"""Deterministic synthetic cohorts, shared by tests, fixtures and bench.py.

Shapes follow the reference's loader and configs:
  * observations grouped by output, n_d = N // D (+1 for the first N % D outputs)
    (ref: dataio/c_experiment.cpp:272-308 appends feature by feature);
  * t ~ U(0, 200) hours, float32, sorted within each output, ~25 % of time stamps shared across
    outputs (labs drawn together => r = 0 off-diagonal pairs);
  * y ~ N(0, 1) float32 (inputs are z-scored, ref: c_experiment.cpp:304-305);
  * theta drawn inside the opt_prior*.json bounds like c_experiment::get_hyp_LMC_SM
    (ref: dataio/c_experiment.cpp:532-564, scripts/opt_prior2.json:9-20).
Generator: numpy Philox keyed by (seed, patient index) -- counter based, platform independent.
"""
import numpy as np

REF_PI = 3.14159265


def _rng(seed, p, stream=0):
    return np.random.Generator(np.random.Philox(key=[(seed * 1_000_003 + p) & 0xFFFFFFFFFFFFFFFF, stream]))


def patient(seed, p, D, N, shared_frac=0.25, interleave=False):
    """Returns (meta int32[N], t float32[N], y float32[N])."""
    g = _rng(seed, p, 0)
    counts = np.full(D, N // D, dtype=np.int64)
    counts[: N % D] += 1
    pool = g.uniform(0.0, 200.0, size=max(8, N // 4)).astype(np.float32)   # shared draw times
    meta, t = [], []
    for d in range(D):
        nd = int(counts[d])
        own = g.uniform(0.0, 200.0, size=nd).astype(np.float32)
        use_shared = g.random(nd) < shared_frac
        pick = pool[g.integers(0, pool.shape[0], size=nd)]
        td = np.where(use_shared, pick, own).astype(np.float32)
        td.sort()
        meta.append(np.full(nd, d, dtype=np.int32))
        t.append(td)
    meta = np.concatenate(meta)
    t = np.concatenate(t)
    y = g.standard_normal(N).astype(np.float32)
    if interleave:   # exercise the library's regrouping (the survey probe used meta[i] = i % D)
        perm = g.permutation(N)
        meta, t, y = meta[perm], t[perm], y[perm]
    return meta, t, y


def num_hyp(kernel_index, Q, D, R):
    if kernel_index == 7:
        return D + Q * (D * R + 2 + D)
    if kernel_index == 8:
        return 1 + 3 * Q
    if kernel_index == 0:
        return 3
    raise ValueError(kernel_index)


def theta(seed, p, kernel_index, Q, D, R, sparse_frac=0.0):
    """One hyper vector in the reference's theta order (log domain where the reference exps)."""
    g = _rng(seed, p, 1)
    if kernel_index == 7:
        ls = np.log(g.uniform(0.15, 0.4, size=D))
        A = g.uniform(-1.5, 1.5, size=Q * D * R) * 0.9 / np.sqrt(Q * R)
        if sparse_frac > 0:
            A[g.random(A.shape[0]) < sparse_frac] = 0.0
        lmu = np.log(1.0 / g.uniform(12.0, 72.0, size=Q))
        lv = np.log(1.0 / (2 * REF_PI * g.uniform(6.0, 72.0, size=Q)))
        lk = np.log(g.uniform(0.1, 0.5, size=Q * D) * 0.1 / Q)
        return np.concatenate([ls, A, lmu, lv, lk])
    if kernel_index == 8:
        ls = np.log(g.uniform(0.15, 0.4, size=1))
        lw = np.log(g.uniform(0.1, 1.0, size=Q) / Q)
        lmu = np.log(1.0 / g.uniform(12.0, 72.0, size=Q))
        lv = np.log(1.0 / (2 * REF_PI * g.uniform(6.0, 72.0, size=Q)))
        return np.concatenate([ls, lw, lmu, lv])
    if kernel_index == 0:
        return np.array([np.log(g.uniform(0.15, 0.4)), np.log(g.uniform(6.0, 72.0)), np.log(g.uniform(0.5, 1.5))])
    raise ValueError(kernel_index)


def cohort(seed, P, D, N, kernel_index=7, Q=5, R=None, interleave=False, first=0):
    """P patients + one theta each. Returns list of (meta, t, y) and theta [P, H]."""
    if R is None:
        R = min(8, D)
    pts = [patient(seed, first + p, D, N, interleave=interleave) for p in range(P)]
    th = np.stack([theta(seed, first + p, kernel_index, Q, D, R) for p in range(P)])
    return pts, th


def hier_gamma_prior(Q, D, R, beta_lam=0.01):
    """c_prior::setup_hier_gamma_prior as flat per-hyper arrays (ref: prior/c_prior.cpp:222-279)."""
    H = D + Q * (D * R + 2 + D)
    flag = np.zeros(H, np.uint8)
    typ = np.full(H, -1, np.int32)
    ex = np.zeros(H, np.uint8)
    p0 = np.zeros(H, np.float32)
    p1 = np.ones(H, np.float32)
    a0, a1 = D, D + Q * D * R
    flag[a0:a1] = 1
    typ[a0:a1] = 1
    ex[a1:a1 + 2 * Q] = 1
    k0 = D + Q * (D * R + 2)
    flag[k0:] = 1
    typ[k0:] = 2
    ex[k0:] = 1
    p1[k0:] = np.float32(beta_lam)
    return flag, typ, ex, p0, p1


def ragged_sizes(seed, P, median=250.0, sigma=1.0, nmin=8, nmax=6000):
    """Observation counts of a heavy-tailed cohort: log-normal around `median` (real cohorts are: the reference's job generator
    buckets patients by N and gives the large ones more resources, ref: scripts/slurm_della.json:6-62,
    medgpc/util/run_exp_generator.py:213-260).  seed 0, P = 300: median 263, eight patients above N = 1400, the largest 5832."""
    g = np.random.Generator(np.random.Philox(key=[seed, 77]))
    return np.clip(np.exp(np.log(median) + sigma * g.standard_normal(P)), nmin, nmax).astype(np.int64)


def ragged_cohort(seed, P, D, kernel_index=7, Q=5, R=None, **kw):
    """Patients of ragged_sizes(seed, P) + one theta each: list of (meta, t, y), theta [P, H], sizes [P]."""
    if R is None:
        R = min(8, D)
    ns = ragged_sizes(seed, P, **kw)
    pts = [patient(seed + 17, p, D, int(ns[p])) for p in range(P)]
    th = np.stack([theta(seed + 17, p, kernel_index, Q, D, R) for p in range(P)])
    return pts, th, ns
