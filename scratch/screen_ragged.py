"""medgp_screen on the largest patients of the heavy-tailed cohort: ms per call (MEDGP_LIB selects the library build)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import medgp_amd
from medgp_amd import synth
D, Q, R = 24, 5, 8
ns = sorted([max(48, int(v)) for v in synth.ragged_sizes(0, 512)], reverse=True)
for lo, hi, ninit in ((0, 4, 200), (4, 24, 200), (24, 88, 200), (88, 344, 100)):
    sel = ns[lo:hi]
    ctx = medgp_amd.Context(7, Q, D, R); ctx.reserve(len(sel), max(sel), 1024)
    ctx.set_patients(np.arange(len(sel)), [synth.patient(9, p, D, n) for p, n in enumerate(sel)])
    th = np.stack([synth.theta(9, k, 7, Q, D, R) for k in range(ninit)])
    ctx.screen(np.arange(len(sel)), th)
    ctx.profile_reset(); ctx.profile_enable(True)
    t0 = time.perf_counter(); ctx.screen(np.arange(len(sel)), th); dt = time.perf_counter() - t0
    pr = {k: round(v[0], 2) for k, v in ctx.profile_read().items() if v[1] > 0}
    print(os.environ.get("MEDGP_LIB", "default")[-22:], "patients", lo, hi, "n", sel[0], sel[-1], "ms", round(dt * 1e3, 1), pr, flush=True)
    ctx.close()
