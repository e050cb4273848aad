"""persistent look-ahead schedule (kernels_cholinv_lp.h) vs one launch per step (kernels_cholinv_la.h): bitwise A/B + timing.
usage: python scratch/lp_ab.py [reps]   (environment read at context creation: MEDGP_LA_PERSIST)"""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import medgp_amd
from medgp_amd import synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
shapes = [(1, 2048, 24), (1, 4096, 64), (4, 2048, 24), (8, 512, 24), (1, 1000, 24), (3, 1500, 8), (1, 130, 2)]
if os.environ.get("LP_SHAPES"):
    shapes = [tuple(int(x) for x in s.split("x")) for s in os.environ["LP_SHAPES"].split(",")]
def make(persist, P, N, D):
    os.environ["MEDGP_LA_PERSIST"] = str(persist)
    Q, R = 5, min(8, D)
    th = np.stack([synth.theta(11, p, 7, Q, D, R) for p in range(P)])
    pts = [synth.patient(11, p, D, N - 37 * p) for p in range(P)]   # ragged sizes
    ctx = medgp_amd.Context(7, Q, D, R); ctx.reserve(P, N, P)
    ctx.set_patients(np.arange(P), pts)
    return ctx, np.stack(th)
bad = 0
for (P, N, D) in shapes:
    res = {}
    for persist in (0, 1):
        ctx, th = make(persist, P, N, D)
        out = [ctx.nlml_grad(np.arange(P), th, True) for _ in range(reps)]
        t0 = time.perf_counter()
        for _ in range(reps): ctx.nlml_grad(np.arange(P), th, True)
        ms = (time.perf_counter() - t0) / reps * 1e3
        ctx.profile_enable(True)
        for _ in range(3): ctx.nlml_grad(np.arange(P), th, True)
        prof = {k: round(v[0] / 3, 3) for k, v in ctx.profile_read().items() if v[1] > 0}
        res[persist] = (out, ms, prof)
        del ctx
    ref = res[0][0][0]
    same = all(np.array_equal(o[0], ref[0]) and np.array_equal(o[1], ref[1]) and np.array_equal(o[2], ref[2]) for p in (0, 1) for o in res[p][0])
    bad += not same
    print(f"P{P} N{N} D{D}: bitwise {'SAME' if same else 'DIFFERENT'} over {2 * reps} evaluations; status {ref[2][:4]}; nlml0 {ref[0][0]!r}")
    if not same:
        for p in (0, 1):
            for i, o in enumerate(res[p][0]):
                if not (np.array_equal(o[0], ref[0]) and np.array_equal(o[1], ref[1])):
                    print("   persist", p, "rep", i, "nlml", o[0][:2], "max |dgrad|", float(np.max(np.abs(o[1] - ref[1]))), "status", o[2][:4]); break
    print(f"   per step: {res[0][1]:.3f} ms  {res[0][2]}")
    print(f"   persist : {res[1][1]:.3f} ms  {res[1][2]}", flush=True)
print("FAILED" if bad else "ALL SAME")
sys.exit(1 if bad else 0)
