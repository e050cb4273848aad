#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/ (run in the BUILD container only;
it reads /root/reference, which does not exist on the GPU box).

1. fastkernel_*.npz  -- outputs of the reference's own Python statement of B_q and k_q
   (/root/reference/medgpc/visualization/fastkernel.py:13-48), imported here.
2. appendixA_*.npz   -- inputs of SURVEY.md Appendix A's driver (gen_appendixA_inputs.cpp, libstdc++
   distributions on mt19937(1234)) together with the nlml the COMPILED REFERENCE printed for them, as
   recorded in SURVEY.md section 8c (the reference cannot be rebuilt under this round's rules: it needs
   <mkl.h> and rapidjson), plus the oracle's own fp64 outputs for regression.
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

# (D, N, Q, R) -> nlml printed by the compiled reference (SURVEY.md section 8c table, 1 thread)
RECORDED = {
    (2, 150, 5, 2): 807.1832958150,
    (2, 256, 5, 2): 1595.6663842416,
    (24, 512, 5, 8): 1740.5962798340,
    (24, 2048, 5, 8): 10746.7599198414,
}
# prior mode 2 (eta = beta_lam = 0.01), D=24 N=512, 8 threads
RECORDED_PRIOR2 = {(24, 512, 5, 8): 2230.1612403014}


def load_bin(p):
    b = open(p, "rb").read()
    D, N, Q, R, H = [int(v) for v in np.frombuffer(b, np.int32, 5)]
    o = 20
    meta = np.frombuffer(b, np.int32, N, o); o += 4 * N
    x = np.frombuffer(b, np.float32, N, o); o += 4 * N
    y = np.frombuffer(b, np.float32, N, o); o += 4 * N
    th = np.frombuffer(b, np.float64, H, o)
    return D, N, Q, R, meta.copy(), x.copy(), y.copy(), th.copy()


def appendix_a():
    with tempfile.TemporaryDirectory() as td:
        exe = os.path.join(td, "gen")
        subprocess.check_call(["g++", "-O2", "-o", exe, os.path.join(HERE, "gen_appendixA_inputs.cpp")])
        for (D, N, Q, R), ref in RECORDED.items():
            p = os.path.join(td, "a.bin")
            subprocess.check_call([exe, str(D), str(N), str(Q), str(R), "0", p])
            D_, N_, Q_, R_, meta, x, y, th = load_bin(p)
            want_grad = N <= 512
            r = O.nlml_grad(7, Q, D, R, meta, x, y, th, flag_grad=want_grad, nthreads=8)
            out = dict(D=D, N=N, Q=Q, R=R, meta=meta, t=x, y=y, theta=th, ref_fp32_nlml=ref,
                       oracle_nlml=r["nlml"], oracle_status=r["status"])
            if want_grad:
                out["oracle_grad"] = r["grad"]
            if (D, N, Q, R) in RECORDED_PRIOR2:
                pr = O.Prior.hier_gamma(Q, D, R, 0.01, 0.01)
                r2 = O.nlml_grad(7, Q, D, R, meta, x, y, th, flag_grad=True, prior=pr, nthreads=8)
                out["ref_fp32_nlml_prior2"] = RECORDED_PRIOR2[(D, N, Q, R)]
                out["oracle_nlml_prior2"] = r2["nlml"]
                out["oracle_grad_prior2"] = r2["grad"]
            np.savez_compressed(os.path.join(HERE, f"appendixA_D{D}_N{N}.npz"), **out)
            print(f"appendixA D={D} N={N}: oracle {r['nlml']:.10f} recorded reference {ref:.10f} "
                  f"rel {abs(r['nlml'] - ref) / ref:.2e}")


def fastkernel():
    # import the single reference file (its package __init__ pulls in seaborn, absent here)
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_fastkernel", "/root/reference/medgpc/visualization/fastkernel.py")
    fk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fk)
    rng = np.random.default_rng(20240601)
    for (Q, D, R) in [(5, 2, 2), (5, 24, 8), (3, 7, 4)]:
        H = D + Q * (D * R + 2 + D)
        hyp = rng.normal(0, 0.7, size=H)
        B = np.stack(fk.compute_B_matrix(Q, D, R, hyp))            # fastkernel.py:13-22
        x = rng.uniform(0, 200, size=(40, 1))
        mu = np.exp(hyp[D + Q * D * R: D + Q * D * R + Q])
        v = np.exp(hyp[D + Q * D * R + Q: D + Q * D * R + 2 * Q])
        # fastkernel's v argument is v^2 (feature_extraction.py:75-77: exp(2 * theta_v))
        resp = np.stack([fk.compute_sm_1d(mu[q], v[q] ** 2, x)[:, 0] for q in range(Q)])   # fastkernel.py:24-48
        np.savez_compressed(os.path.join(HERE, f"fastkernel_Q{Q}_D{D}_R{R}.npz"), Q=Q, D=D, R=R, hyp=hyp, B=B,
                            x=x[:, 0], mu=mu, v=v, resp=resp)
        print(f"fastkernel Q={Q} D={D} R={R}: B {B.shape}, resp {resp.shape}")


if __name__ == "__main__":
    appendix_a()
    fastkernel()
