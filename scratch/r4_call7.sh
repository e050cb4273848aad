#!/bin/bash
# round 4, GPU call 7: GPU suite incl. the two-launch tuned kernels for 9..16 components, bench line, profile set of the final device sources
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c7
O=gpurun_out/r4c7
(time python3 -m pytest tests -m gpu -q) > $O/pytest.log 2>&1
tail -4 $O/pytest.log; grep -E "^FAILED|^ERROR" $O/pytest.log | head
python3 - > $O/q12.log 2>&1 <<'PY'
import sys, os, time, numpy as np
sys.path.insert(0, os.getcwd())
import medgp_amd
from medgp_amd import synth
for Q in (8, 12, 17):
    D, N, R, P = 24, 512, 8, 256
    pts, th = synth.cohort(5, 8, D, N, Q=Q, R=R)
    ctx = medgp_amd.Context(7, Q, D, R); ctx.reserve(P, N, P); ctx.set_patients(np.arange(P), [pts[s % 8] for s in range(P)])
    thp = np.stack([th[s % 8] for s in range(P)])
    for _ in range(2): ctx.nlml_grad(np.arange(P), thp, True)
    ctx.profile_enable(True)
    for _ in range(3): ctx.nlml_grad(np.arange(P), thp, True)
    print("Q", Q, {k: round(v[0] / 3, 3) for k, v in ctx.profile_read().items() if v[1] > 0})
    ctx.close()
PY
cat $O/q12.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.log 2>&1
tail -1 $O/bench_driver_cmd.log | cut -c1-300
bash scratch/gpurun_prof.sh r04 2>&1 | tail -3
