"""How much could an fp32 factorisation pay for the nlml-only screening batch (ref: main_one_train.cpp:228-253: 1000 random
hyper vectors of ONE patient, flag_grad = false)?  Times, on the same GPU:
  (a) this library, fp64: k_prep + k_assemble + k_cholinv (L, z, log det; no inverse) + k_epilogue for 1000 x N
  (b) the vendor batched Cholesky (torch.linalg.cholesky -> rocSOLVER/MAGMA) in fp32 and in fp64 on 1000 SPD matrices of
      the same size -- the factorisation alone, no assembly, no solve: the fp32 / fp64 ratio a tuned library reaches here
torch is used as a measuring stick only (scratch experiment, not product).  usage: python scratch/fp32_potrf_time.py [N] [P]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import medgp_amd
from medgp_amd import synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
P = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
D, Q, R = 24, 5, 8
m, t, y = synth.patient(2024, 0, D, N)
th = np.stack([synth.theta(2024, s, 7, Q, D, R) for s in range(P)])
ctx = medgp_amd.Context(7, Q, D, R); ctx.reserve(1, N, P); ctx.set_patient(0, m, t, y)
slots = np.zeros(P, dtype=np.int32)
for _ in range(2): nl, _, st = ctx.nlml_grad(slots, th, False)
assert np.all(st == 0)
ctx.profile_enable(True)
reps = 5
for _ in range(reps): ctx.nlml_grad(slots, th, False)
prof = {k: round(v[0] / reps, 3) for k, v in ctx.profile_read().items() if v[1] > 0}
print(f"ours fp64, {P} x N={N} nlml-only: kernels {prof}  sum {sum(prof.values()):.3f} ms")

dev = torch.device("cuda")
g = torch.Generator(device="cpu").manual_seed(0)
B = torch.randn(P, N, N, generator=g, dtype=torch.float64)
A64 = (B @ B.transpose(1, 2) / N + torch.eye(N, dtype=torch.float64)).to(dev)
A32 = A64.float()
def tm(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
t64 = tm(lambda: torch.linalg.cholesky(A64))
t32 = tm(lambda: torch.linalg.cholesky(A32))
L64 = torch.linalg.cholesky(A64); L32 = torch.linalg.cholesky(A32)
ld64 = torch.log(torch.diagonal(L64, dim1=1, dim2=2)).sum(1)
ld32 = torch.log(torch.diagonal(L32, dim1=1, dim2=2).double()).sum(1)
print(f"vendor batched potrf, {P} x {N}x{N}: fp64 {t64:.3f} ms, fp32 {t32:.3f} ms (ratio {t64 / t32:.2f}); "
      f"log det rel. error of the fp32 factor: max {((ld32 - ld64).abs() / ld64.abs()).max().item():.2e}")
