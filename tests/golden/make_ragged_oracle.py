#!/usr/bin/env python3
"""Oracle outputs for the LARGE patients of the heavy-tailed cohort of tests/test_ragged_gpu.py (synth.ragged_cohort(0, 300, 24)):
the CPU oracle needs minutes for them (N = 5832: ~2 min on 8 threads), so the GPU test reads them from
tests/golden/ragged_cohort_large.npz and runs the oracle live only for the small patients.  Inputs are NOT stored: they are
regenerated from the seed by medgp_amd.synth (counter-based Philox, platform independent).
Run here or anywhere the oracle builds: python tests/golden/make_ragged_oracle.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from medgp_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

SEED, P, D, Q, R = 0, 300, 24, 5, 8
LARGE = 1200   # patients above this many observations come from the fixture


def main():
    pts, th, ns = synth.ragged_cohort(SEED, P, D, 7, Q, R)
    prior = O.Prior.hier_gamma(Q, D, R, 0.01, 0.01)
    idx = [p for p in range(P) if ns[p] > LARGE]
    out = dict(seed=SEED, P=P, D=D, Q=Q, R=R, large=LARGE, index=np.array(idx), n=ns[idx])
    nl, gr, st = [], [], []
    for p in idx:
        m, t, y = pts[p]
        r = O.nlml_grad(7, Q, D, R, m, t, y, th[p], prior=prior, nthreads=os.cpu_count())
        print(p, ns[p], r["status"], r["nlml"], flush=True)
        nl.append(r["nlml"]); gr.append(r["grad"]); st.append(r["status"])
    out.update(nlml=np.array(nl), grad=np.stack(gr), status=np.array(st))
    np.savez_compressed(os.path.join(HERE, "ragged_cohort_large.npz"), **out)


if __name__ == "__main__":
    main()
