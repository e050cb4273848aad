#!/bin/bash
# round 4, GPU call 6: PCIe-inclusive rate of the host-pointer API at the headline shape; larger cohort imputation test (robustness)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c6
O=gpurun_out/r4c6
python3 - > $O/pcie.log 2>&1 <<'PY'
import sys, os, time, numpy as np
sys.path.insert(0, os.getcwd())
import medgp_amd
from medgp_amd import synth
D, N, Q, R, P = 24, 512, 5, 8, 512
pts, th = synth.cohort(2024, P, D, N, Q=Q, R=R)
ctx = medgp_amd.Context(7, Q, D, R); ctx.reserve(P, N, P); ctx.set_patients(np.arange(P), pts)
ctx.set_prior(-1, *synth.hier_gamma_prior(Q, D, R, 0.01))
sl = np.arange(P)
for _ in range(3): ctx.nlml_grad(sl, th, True)
t0 = time.perf_counter()
for _ in range(20): ctx.nlml_grad(sl, th, True)
dt = (time.perf_counter() - t0) / 20
print(f"host-pointer API, pageable numpy arrays: {1e3*dt:.3f} ms per 512-patient step = {P/dt:.0f} evals/s")
H = ctx.H
thp = ctx.pinned((P, H), np.float64); thp[:] = th
nl = ctx.pinned((P,), np.float64); g = ctx.pinned((P, H), np.float64); st = ctx.pinned((P,), np.int32)
for _ in range(3): ctx.nlml_grad_async(0, sl, thp, True, nl, g, st); ctx.wait(0)
t0 = time.perf_counter()
for _ in range(20): ctx.nlml_grad_async(0, sl, thp, True, nl, g, st); ctx.wait(0)
dt = (time.perf_counter() - t0) / 20
print(f"async lane on pinned arrays (medgp_nlml_grad_async + medgp_wait): {1e3*dt:.3f} ms per step = {P/dt:.0f} evals/s")
PY
cat $O/pcie.log
python3 scratch/impute_cohort_time.py 256 300 4 > $O/impute256.log 2>&1; grep -v amdgpu $O/impute256.log | cut -c1-300
