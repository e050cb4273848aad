"""medgp_screen, one lane against two (MEDGP_SCREEN_LANES): P patients of N = 512 (D = 24), 1000 hyper vectors each, max_batch 1024 --
the trainer's screening call.  python scratch/screen_lanes.py [P]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import medgp_amd
from medgp_amd import synth
D, N, Q, R, NI = 24, 512, 5, 8, 1000
P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
th = np.stack([synth.theta(2026, s, 7, Q, D, R) for s in range(NI)])
pts = [synth.patient(2026, p, D, N) for p in range(P)]
res = {}
for rnd in range(3):
    for lanes in ("1", "2"):
        os.environ["MEDGP_SCREEN_LANES"] = lanes
        c = medgp_amd.Context(7, Q, D, R); c.reserve(P, N, 1024)
        c.set_patients(np.arange(P), pts)
        c.reserve_plan([N] * P, NI)
        nl, st = c.screen(np.arange(P), th)
        t0 = time.perf_counter()
        for _ in range(3): nl, st = c.screen(np.arange(P), th)
        dt = (time.perf_counter() - t0) / 3
        assert (st == 0).all()
        res.setdefault(lanes, nl)
        assert np.array_equal(res[lanes], nl)
        print(f"round {rnd} lanes {lanes}: {1e3 * dt / P:.3f} ms per 1000 evaluations, {P * NI / dt / 1e3:.1f} k evals/s; arenas {c.alloc_stats()[2] / 2**30:.2f} GB", flush=True)
        c.close()
print("one lane == two lanes bit for bit:", np.array_equal(res["1"], res["2"]))
