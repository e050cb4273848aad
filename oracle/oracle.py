"""ctypes binding of the CPU oracle (oracle/libmedgp_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (medgp_amd/) must never import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libmedgp_oracle.so")

KERNEL_SE, KERNEL_LMC_SM, KERNEL_SM = 0, 7, 8
REF_PI = 3.14159265  # ref: medgpc/src/util/global_settings.h:6
GRAD_PER_HYPER, GRAD_BLOCKED = 0, 1


def build(force=False):
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(
            os.path.join(_HERE, "medgp_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _LIB


_lib = None
os.environ.setdefault("OMP_WAIT_POLICY", "passive")   # no spinning when the box has fewer cores than threads


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB)
        _lib.medgp_oracle_sm_k.restype = C.c_double
        _lib.medgp_oracle_sm_k.argtypes = [C.c_double] * 4
    return _lib


def _p(a, ty):
    return None if a is None else a.ctypes.data_as(C.POINTER(ty))


def num_hyp(kidx, Q, D, R):
    return lib().medgp_oracle_num_hyp(int(kidx), int(Q), int(D), int(R))


def num_lik(kidx, D):
    return lib().medgp_oracle_num_lik(int(kidx), int(D))


def coregional(Q, D, R, theta_cov):
    Q, D, R = int(Q), int(D), int(R)
    theta_cov = np.ascontiguousarray(theta_cov, dtype=np.float64)
    B = np.empty((Q, D, D), dtype=np.float64)
    lib().medgp_oracle_lmc_coregional(Q, D, R, _p(theta_cov, C.c_double), _p(B, C.c_double))
    return B


def sm_k(rsq, mu, v, pi=REF_PI):
    return lib().medgp_oracle_sm_k(float(rsq), float(mu), float(v), float(pi))


def gram(kidx, Q, D, R, meta, t, theta, pi=REF_PI):
    kidx, Q, D, R = int(kidx), int(Q), int(D), int(R)
    t = np.ascontiguousarray(t, dtype=np.float32)
    n = t.shape[0]
    meta = None if meta is None else np.ascontiguousarray(meta, dtype=np.int32)
    theta = np.ascontiguousarray(theta, dtype=np.float64)
    K = np.empty((n, n), dtype=np.float64)
    ok = lib().medgp_oracle_gram(kidx, Q, D, R, C.c_double(pi), n, _p(meta, C.c_int32), _p(t, C.c_float),
                                 _p(theta, C.c_double), _p(K, C.c_double))
    assert ok
    return K


class Prior:
    """Per-hyper prior descriptor in theta order (mirrors c_prior's flag/type/exp/fix_param vectors)."""

    def __init__(self, H):
        self.flag = np.zeros(H, dtype=np.uint8)
        self.type = np.full(H, -1, dtype=np.int32)
        self.exp = np.zeros(H, dtype=np.uint8)
        self.p0 = np.zeros(H, dtype=np.float32)
        self.p1 = np.ones(H, dtype=np.float32)

    @staticmethod
    def hier_gamma(Q, D, R, eta=0.01, beta_lam=0.01):
        """c_prior::setup_hier_gamma_prior (ref: prior/c_prior.cpp:222-279) for LMC-SM: A ~ Normal(0, 1),
        kappa ~ Laplace(0, beta_lam) with exp chain rule; sigma, mu, v: none."""
        H = D + Q * (D * R + 2 + D)
        p = Prior(H)
        a0, a1 = D, D + Q * D * R
        p.flag[a0:a1] = 1
        p.type[a0:a1] = 1
        p.p0[a0:a1] = 0.0
        p.p1[a0:a1] = 1.0
        p.exp[a1:a1 + 2 * Q] = 1
        k0 = D + Q * (D * R + 2)
        p.flag[k0:] = 1
        p.type[k0:] = 2
        p.exp[k0:] = 1
        p.p0[k0:] = 0.0
        p.p1[k0:] = np.float32(beta_lam)
        return p


def nlml_grad(kidx, Q, D, R, meta, t, y, theta, flag_grad=True, grad_mode=GRAD_BLOCKED, nthreads=1,
              prior=None, pi=REF_PI, want_alpha=False, want_linv=False):
    kidx, Q, D, R = int(kidx), int(Q), int(D), int(R)
    t = np.ascontiguousarray(t, dtype=np.float32)
    y = np.ascontiguousarray(y, dtype=np.float32)
    n = t.shape[0]
    meta = None if meta is None else np.ascontiguousarray(meta, dtype=np.int32)
    theta = np.ascontiguousarray(theta, dtype=np.float64)
    H = num_hyp(kidx, Q, D, R)
    assert theta.shape[0] == H, (theta.shape, H)
    nlml = C.c_double(float("nan"))
    beta = C.c_double(float("nan"))
    status = C.c_int32(-99)
    grad = np.full(H, np.nan) if flag_grad else None
    alpha = np.empty(n) if want_alpha else None
    linv = np.empty((n, n)) if want_linv else None
    pr = prior
    ok = lib().medgp_oracle_nlml_grad(
        kidx, Q, D, R, C.c_double(pi), n, _p(meta, C.c_int32), _p(t, C.c_float), _p(y, C.c_float),
        _p(theta, C.c_double), int(bool(flag_grad)), int(grad_mode), int(nthreads),
        _p(pr.flag, C.c_uint8) if pr else None, _p(pr.type, C.c_int32) if pr else None,
        _p(pr.exp, C.c_uint8) if pr else None, _p(pr.p0, C.c_float) if pr else None,
        _p(pr.p1, C.c_float) if pr else None,
        C.byref(nlml), _p(grad, C.c_double), _p(alpha, C.c_double), _p(linv, C.c_double), C.byref(beta),
        C.byref(status))
    out = {"ok": bool(ok), "nlml": nlml.value, "grad": grad, "status": status.value, "beta": beta.value}
    if want_alpha:
        out["alpha"] = alpha
    if want_linv:
        out["linv"] = linv
    return out


def fit_predict(kidx, Q, D, R, meta, t, y, theta, meta2, t2, pi=REF_PI):
    kidx, Q, D, R = int(kidx), int(Q), int(D), int(R)
    t = np.ascontiguousarray(t, dtype=np.float32)
    y = np.ascontiguousarray(y, dtype=np.float32)
    t2 = np.ascontiguousarray(t2, dtype=np.float32)
    meta = None if meta is None else np.ascontiguousarray(meta, dtype=np.int32)
    meta2 = None if meta2 is None else np.ascontiguousarray(meta2, dtype=np.int32)
    theta = np.ascontiguousarray(theta, dtype=np.float64)
    ns = t2.shape[0]
    mean = np.empty(ns)
    var = np.empty(ns)
    status = C.c_int32(-99)
    ok = lib().medgp_oracle_fit_predict(kidx, Q, D, R, C.c_double(pi), t.shape[0], _p(meta, C.c_int32),
                                        _p(t, C.c_float), _p(y, C.c_float), _p(theta, C.c_double), ns,
                                        _p(meta2, C.c_int32), _p(t2, C.c_float), _p(mean, C.c_double),
                                        _p(var, C.c_double), C.byref(status))
    return {"ok": bool(ok), "mean": mean, "var": var, "status": status.value}
