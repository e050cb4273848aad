"""per-kernel HIP-event totals of the ragged call, classes on separate streams vs back to back"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import medgp_amd
from medgp_amd import synth
P, D, Q, R = 300, 24, 5, 8
pts, th, ns = synth.ragged_cohort(0, P, D, 7, Q, R)
for env in ({}, {"MEDGP_CLASS_STREAMS": "0"}):
    os.environ.pop("MEDGP_CLASS_STREAMS", None); os.environ.update(env)
    ctx = medgp_amd.Context(7, Q, D, R); ctx.reserve(P, int(ns.max()), P)
    ctx.set_patients(np.arange(P), pts); ctx.set_prior(-1, *synth.hier_gamma_prior(Q, D, R, 0.01))
    sl = np.arange(P)
    for _ in range(2): ctx.nlml_grad(sl, th, True)
    t0 = time.perf_counter()
    for _ in range(5): ctx.nlml_grad(sl, th, True)
    wall = (time.perf_counter() - t0) / 5 * 1e3
    ctx.profile_reset(); ctx.profile_enable(True)
    for _ in range(3): ctx.nlml_grad(sl, th, True)
    prof = {k: (round(v[0] / 3, 3), v[1] // 3) for k, v in ctx.profile_read().items() if v[1] > 0}
    ctx.profile_enable(False)
    print(env, "wall", round(wall, 3), "kernel ms (launches) per call:", prof, "sum", round(sum(v[0] for v in prof.values()), 3), flush=True)
    # each class alone
    if not env:
        nb = (ns + 63) // 64
        bucket = np.ceil(np.log2(np.maximum(nb, 1))).astype(int)
        for bk in sorted(set(bucket.tolist()), reverse=True):
            s2 = sl[bucket == bk]
            for _ in range(2): ctx.nlml_grad(s2, th[s2], True)
            ctx.profile_reset(); ctx.profile_enable(True)
            for _ in range(3): ctx.nlml_grad(s2, th[s2], True)
            pr = {k: round(v[0] / 3, 3) for k, v in ctx.profile_read().items() if v[1] > 0}
            ctx.profile_enable(False)
            print("  class 2^%d: %3d entries" % (bk, len(s2)), pr, flush=True)
    ctx.close()
