"""cost of the per-launch HIP events (ctx.profile_enable) inside a timed region: 300 steps of the headline shape, events on / off, interleaved"""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import medgp_amd
from medgp_amd import synth
P, N, D, Q, R = 512, 512, 24, 5, 8
pts, th = synth.cohort(11, 8, D, N, Q=Q, R=R)
ctx = medgp_amd.Context(7, Q, D, R); ctx.reserve(P, N, P)
ctx.set_patients(np.arange(P), [pts[s % 8] for s in range(P)])
ctx.set_prior(-1, *synth.hier_gamma_prior(Q, D, R, 0.01))
H = ctx.H
dev = torch.device("cuda:0")
theta_d = torch.from_numpy(np.stack([th[s % 8] for s in range(P)])).to(dev)
nl = torch.empty(P, dtype=torch.float64, device=dev); gr = torch.empty((P, H), dtype=torch.float64, device=dev); st = torch.empty(P, dtype=torch.int32, device=dev)
slots = np.arange(P, dtype=np.int32)
def step(): ctx.nlml_grad_device(slots, theta_d.data_ptr(), True, nl.data_ptr(), gr.data_ptr(), st.data_ptr())
for _ in range(300): step()
ctx.synchronize()
for rnd in range(3):
    for on in (False, True):
        ctx.profile_reset(); ctx.profile_enable(on)
        ctx.synchronize(); t0 = time.perf_counter()
        for _ in range(300): step()
        ctx.synchronize(); dt = (time.perf_counter() - t0) / 300 * 1e3
        ctx.profile_enable(False)
        print(f"round {rnd} events {'on ' if on else 'off'}: {dt:.4f} ms per step", flush=True)
