"""Timing of ONE ragged call (synth.ragged_cohort(seed, P, D)): default plan (size classes on separate streams), classes back to back
(MEDGP_CLASS_STREAMS=0), the rounds 1-4 behaviour (MEDGP_NO_CLASSES=1: one route for the whole call from its largest entry), and the
size classes as separate calls.  usage: python scratch/ragged_time.py [P] [D] [reps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import medgp_amd
from medgp_amd import synth

P = int(sys.argv[1]) if len(sys.argv) > 1 else 300
D = int(sys.argv[2]) if len(sys.argv) > 2 else 24
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 5
Q, R = 5, min(8, D)
pts, th, ns = synth.ragged_cohort(0, P, D, 7, Q, R)
falg = float(sum(n ** 3 + 6 * n ** 2 + 80 * Q * n * (n + 1) / 2 for n in ns.astype(np.float64)))


def make():
    ctx = medgp_amd.Context(7, Q, D, R)
    ctx.reserve(P, int(ns.max()), P)
    ctx.set_patients(np.arange(P), pts)
    ctx.set_prior(-1, *synth.hier_gamma_prior(Q, D, R, 0.01))
    return ctx


def timed(ctx, slots, reps):
    ctx.nlml_grad(slots, th[slots], True)
    t0 = time.perf_counter()
    for _ in range(reps):
        out = ctx.nlml_grad(slots, th[slots], True)
    return (time.perf_counter() - t0) / reps * 1e3, out


allp = np.arange(P)
res = {}
for name, env, reps in (("default", {}, REPS), ("one_stream", {"MEDGP_CLASS_STREAMS": "0"}, REPS), ("no_classes", {"MEDGP_NO_CLASSES": "1"}, 1)):
    for k in ("MEDGP_CLASS_STREAMS", "MEDGP_NO_CLASSES"):
        os.environ.pop(k, None)
    os.environ.update(env)
    ctx = make()
    ms, out = timed(ctx, allp, reps)
    res[name] = out
    print(f"{name:12s} {ms:9.3f} ms/call  {P / ms * 1e3:9.0f} evals/s  {falg / ms / 1e9 / 78.6:6.3f} of fp64 peak   plan {ctx.last_plan()}", flush=True)
    if name == "default":
        plan = ctx.last_plan()
        nb = (ns + 63) // 64
        bucket = np.ceil(np.log2(np.maximum(nb, 1))).astype(int)
        tot = 0.0
        for bk in sorted(set(bucket.tolist()), reverse=True):
            sl = allp[bucket == bk]
            ms1, _ = timed(ctx, sl, REPS)
            tot += ms1
            print(f"   class nb<=2^{bk}: {len(sl):4d} entries {ms1:9.3f} ms  plan {ctx.last_plan()}", flush=True)
        print(f"   sum of the classes as separate calls: {tot:9.3f} ms")
    ctx.close()
for k in ("MEDGP_CLASS_STREAMS", "MEDGP_NO_CLASSES"):
    os.environ.pop(k, None)
a, b = res["default"], res["one_stream"]
print("default == one_stream bit for bit:", all(np.array_equal(x, y) for x, y in zip(a, b)))
c = res["no_classes"]
print("default vs no_classes: max rel nlml diff", float(np.max(np.abs(a[0] - c[0]) / np.abs(c[0]))), "status equal", np.array_equal(a[2], c[2]))
