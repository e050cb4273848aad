"""What an fp32 factorisation + fp64 iterative refinement would cost in accuracy (CPU experiment, numpy/LAPACK).

For the bench patient of a config: K (fp64, oracle Gram + noise), then
  fp64 : L = chol(K), logdet, alpha, W = K^-1 - alpha alpha^T                              (what the build computes)
  fp32 : L32 = spotrf(float(K)); logdet32 = sum log L32_ii (added in fp64); alpha32 by spotrs;
         IR: r = y - K alpha (fp64), alpha += spotrs(r), k steps; W32 = float(L32^-T L32^-1) - alpha alpha^T
and the relative errors of nlml and of the sigma / mu / v gradient entries against the fp64 values.
usage: python scratch/fp32_ir_accuracy.py [D N]    (test infrastructure: uses oracle/)
"""
import os, sys
import numpy as np
import scipy.linalg as sl

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle
from medgp_amd import synth

PI = oracle.REF_PI


def run(D, N, Q=5, R=8, seed=2024):
    meta, t, y = synth.patient(seed, 0, D, N)
    th = synth.theta(seed, 0, 7, Q, D, R)
    ref = oracle.nlml_grad(7, Q, D, R, meta, t, y, th, flag_grad=True, nthreads=8)
    K = oracle.gram(7, Q, D, R, meta, t, th)
    sig2 = np.exp(2 * th[:D])
    if abs(K[0, 0] - (oracle.gram(7, Q, D, R, meta, t, th)[0, 0])) == 0 and True:
        pass
    yv = y.astype(np.float64)
    n = len(yv)

    def nlml_of(logdet, quad):
        return 0.5 * quad + logdet + 0.5 * n * np.log(2 * PI)

    # does the oracle Gram carry the noise diagonal?  decide by matching the oracle's nlml
    for add in (0.0, 1.0):
        K1 = K + add * np.diag(sig2[meta])
        L = np.linalg.cholesky(K1)
        al = sl.cho_solve((L, True), yv)
        v = nlml_of(np.log(np.diag(L)).sum(), yv @ al)
        if abs(v - ref["nlml"]) <= 1e-9 * abs(v):
            K = K1
            break
    else:
        raise SystemExit("cannot reproduce the oracle nlml")
    L = np.linalg.cholesky(K)
    alpha = sl.cho_solve((L, True), yv)
    nl64 = nlml_of(np.log(np.diag(L)).sum(), yv @ alpha)
    Li = sl.solve_triangular(L, np.eye(n), lower=True)
    W64 = Li.T @ Li - np.outer(alpha, alpha)
    print(f"D={D} N={N}: cond(K) ~ {np.linalg.cond(K):.2e}, nlml fp64 {nl64:.10f} (oracle {ref['nlml']:.10f})")

    K32 = K.astype(np.float32)
    L32 = sl.cholesky(K32, lower=True)
    assert L32.dtype == np.float32
    logdet32 = np.log(np.diag(L32).astype(np.float64)).sum()
    a = sl.cho_solve((L32, True), y.astype(np.float32)).astype(np.float64)
    print(f"  log det: fp32 factor rel err {abs(logdet32 - np.log(np.diag(L)).sum()) / abs(np.log(np.diag(L)).sum()):.2e}  (not refinable without an fp64 factor)")
    for it in range(4):
        q = yv @ a
        print(f"  IR step {it}: nlml rel err {abs(nlml_of(logdet32, q) - nl64) / abs(nl64):.2e}, quad rel err {abs(q - yv @ alpha) / abs(yv @ alpha):.2e}, |alpha err|/|alpha| {np.abs(a - alpha).max() / np.abs(alpha).max():.2e}")
        r = yv - K @ a
        a = a + sl.cho_solve((L32, True), r.astype(np.float32)).astype(np.float64)
    Li32 = sl.solve_triangular(L32, np.eye(n, dtype=np.float32), lower=True)
    W32 = (Li32.T @ Li32).astype(np.float64) - np.outer(a, a)

    # gradient entries that need W: noise (a15) and the mu / v entries (a16)
    B = oracle.coregional(Q, D, R, th[D:])
    o_mu = D + Q * D * R
    mu = np.exp(th[o_mu:o_mu + Q]); v = np.exp(th[o_mu + Q:o_mu + 2 * Q])
    tt = t.astype(np.float64)
    r = np.abs(tt[:, None] - tt[None, :]); rsq = r * r

    def grads(W):
        g = [sig2[d] * np.trace(W[np.ix_(meta == d, meta == d)]) for d in range(D)]
        for q in range(Q):
            Bm = B[q][np.ix_(meta, meta)]
            E = np.exp(-2 * (PI * v[q]) ** 2 * rsq)
            g.append(0.5 * np.sum(W * Bm * (-(2 * PI * r * mu[q]) * np.sin(2 * PI * r * mu[q]) * E)))
        for q in range(Q):
            Bm = B[q][np.ix_(meta, meta)]
            k = np.cos(2 * PI * r * mu[q]) * np.exp(-2 * (PI * v[q]) ** 2 * rsq)
            g.append(0.5 * np.sum(W * Bm * (-4 * (PI * v[q]) ** 2 * rsq * k)))
        return np.array(g)

    g64, g32 = grads(W64), grads(W32)
    gref = np.concatenate([ref["grad"][:D], ref["grad"][o_mu:o_mu + 2 * Q]])
    scale = np.maximum(np.abs(g64), 1e-3 * np.abs(g64).max())
    print(f"  sanity: numpy fp64 gradient vs oracle, max rel {np.max(np.abs(g64 - gref) / scale):.2e}")
    e = np.abs(g32 - g64) / scale
    print(f"  gradient (sigma, mu, v entries) with the fp32 inverse: median rel err {np.median(e):.2e}, max {e.max():.2e}   (tolerance 1e-6)")


if __name__ == "__main__":
    if len(sys.argv) > 2:
        run(int(sys.argv[1]), int(sys.argv[2]))
    else:
        run(24, 512)
        run(24, 2048)
        run(64, 4096)
