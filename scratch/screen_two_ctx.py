"""The "cheap alternative" of the round-5 verdict (item 2), measured without building it: would running the assembly of one screening
chunk beside the factorisation of another (two streams) pay?  Two CONTEXTS (independent buffers, each on its own stream) screen the
same patient from two host threads at once; their kernels interleave freely on the device -- k_assemble_t (fp64 VALU bound) of one
beside k_cholinv (MFMA + latency bound) of the other.  Compared with ONE context doing all the work back to back.
python scratch/screen_two_ctx.py"""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import medgp_amd
from medgp_amd import synth
D, N, Q, R, P, REPS = 24, 512, 5, 8, 1000, 40
th = np.stack([synth.theta(2026, s, 7, Q, D, R) for s in range(P)])
def mk():
    c = medgp_amd.Context(7, Q, D, R); c.reserve(1, N, 1024)
    c.set_patient(0, *synth.patient(2026, 0, D, N))
    c.screen(np.zeros(1, np.int32), th)
    return c
a, b = mk(), mk()
def run(c, n):
    for _ in range(n): c.screen(np.zeros(1, np.int32), th)
for rnd in range(3):
    t0 = time.perf_counter(); run(a, 2 * REPS); t1 = time.perf_counter() - t0
    ts = [threading.Thread(target=run, args=(c, REPS)) for c in (a, b)]
    t0 = time.perf_counter(); [t.start() for t in ts]; [t.join() for t in ts]; t2 = time.perf_counter() - t0
    print(f"round {rnd}: one context {2 * REPS * P / t1 / 1e3:.1f} k evals/s ({1e3 * t1 / (2 * REPS):.3f} ms per 1000); two contexts at once {2 * REPS * P / t2 / 1e3:.1f} k evals/s ({1e3 * t2 / (2 * REPS):.3f} ms per 1000): x{t1 / t2:.3f}", flush=True)
