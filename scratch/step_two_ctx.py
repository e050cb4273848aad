"""Would running the gradient kernel (k_wgrad) of one half of the batch beside the factorisation (k_cholinv) of the other half pay?
One k_cholinv<4,4> workgroup (82 KB LDS, 256 VGPRs x 4 waves) and two k_wgrad workgroups (41 KB, 128 VGPRs) fit one CU together.
Measured without building it: TWO contexts of 256 patients each (own streams) step concurrently from two host threads, against ONE
context stepping all 512.  python scratch/step_two_ctx.py   (MEDGP_CHOLINV_NW=44 keeps the 4-wave shape for the halves)"""
import os, sys, threading, time
import numpy as np
os.environ.setdefault("MEDGP_CHOLINV_NW", "44")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import medgp_amd
from medgp_amd import synth
D, N, Q, R = 24, 512, 5, 8
H = synth.num_hyp(7, Q, D, R)
dev = torch.device("cuda", 0)
def mk(first, P):
    c = medgp_amd.Context(7, Q, D, R, device=0); c.reserve(P, N, P)
    c.set_patients(np.arange(P), [synth.patient(2024, first + s, D, N) for s in range(P)])
    c.set_prior(-1, *synth.hier_gamma_prior(Q, D, R, 0.01))
    th = torch.from_numpy(np.stack([synth.theta(2024, first + s, 7, Q, D, R) for s in range(P)])).to(dev)
    out = (torch.empty(P, dtype=torch.float64, device=dev), torch.empty((P, H), dtype=torch.float64, device=dev), torch.empty(P, dtype=torch.int32, device=dev))
    sl = np.arange(P, dtype=np.int32)
    def step(): c.nlml_grad_device(sl, th.data_ptr(), 1, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr())
    for _ in range(3): step()
    c.synchronize()
    return c, step, out
one = mk(0, 512); a = mk(0, 256); b = mk(256, 256)
def run(ctx, step, n, per_sync):
    for i in range(n):
        step()
        if (i + 1) % per_sync == 0: ctx.synchronize()
    ctx.synchronize()
K = 200
for per_sync in (1, 4):
    for rnd in range(2):
        t0 = time.perf_counter(); run(one[0], one[1], K, per_sync); t1 = time.perf_counter() - t0
        ts = [threading.Thread(target=run, args=(c[0], c[1], K, per_sync)) for c in (a, b)]
        t0 = time.perf_counter(); [t.start() for t in ts]; [t.join() for t in ts]; t2 = time.perf_counter() - t0
        print(f"sync every {per_sync} step(s), round {rnd}: one context of 512: {512 * K / t1 / 1e3:.1f} k evals/s ({1e3 * t1 / K:.3f} ms per 512); two contexts of 256 at once: {512 * K / t2 / 1e3:.1f} k evals/s ({1e3 * t2 / K:.3f} ms per 512): x{t1 / t2:.3f}", flush=True)
assert bool((one[2][2] >= 0).all()) and bool((a[2][2] >= 0).all()) and bool((b[2][2] >= 0).all())
print("bits of the halves == bits of the whole:", bool(torch.equal(one[2][0][:256], a[2][0])) and bool(torch.equal(one[2][1][256:], b[2][1])))
