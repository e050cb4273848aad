"""medgp_train on a heavy-tailed cohort (synth.ragged_sizes): continuous admission + per-entry scheduling together.
usage: python scratch/train_ragged.py [P] [ninit] [top_iter] [resident] [extra trainer args...]"""
import os
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from medgp_amd import synth
from medgp_amd.synth_experiment import make_experiment

P = int(sys.argv[1]) if len(sys.argv) > 1 else 300
NINIT = int(sys.argv[2]) if len(sys.argv) > 2 else 50
TOP = int(sys.argv[3]) if len(sys.argv) > 3 else 3
RES = sys.argv[4] if len(sys.argv) > 4 else "256"
ns = [max(48, int(n)) for n in synth.ragged_sizes(0, P)]          # (at least two observations per output at D = 24)
host = os.environ.get("MEDGP_HOST_DIR") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "medgp_amd", "host")
tmp = tempfile.mkdtemp(prefix="medgp_ragged_")
pans = [f"P{k:05d}" for k in range(P)]
ex = make_experiment(os.path.join(tmp, "train"), pans, D=24, Q=5, R=8, N=ns, feature_index=tuple(range(24)), seed=78,
                     opt=dict(random_init_num=NINIT, top_iteration_num=TOP, iteration_num_per_update=30))
plist = os.path.join(tmp, "pans.txt")
open(plist, "w").write("\n".join(pans) + "\n")
f_alg = float(sum(n ** 3 + 6.0 * n ** 2 + 80 * 5 * n * (n + 1) / 2 for n in map(float, ns)))
t0 = time.perf_counter()
r = subprocess.run([os.path.join(host, "medgp_train"), "--cfg", ex["cfg"], "--pan-list", plist, "--resident", RES] + sys.argv[5:],
                   capture_output=True, text=True, timeout=3000)
wall = time.perf_counter() - t0
print("rc", r.returncode, "process wall", round(wall, 2), "s; n max", max(ns), "median", int(np.median(ns)), "mean F_alg per evaluation", f_alg / P)
for ln in r.stdout.splitlines():
    if ln.startswith(("INFO: lock-step", "INFO: continuous", "INFO: gradient", "optimization finished", "ERROR", "Finish all")) or "merged" in ln:
        print(ln)
flags = [open(os.path.join(ex["dirs"]["train"], f"train_flag_{p}.txt")).read().strip() for p in pans]
print("flags 1:", flags.count("1"), "0:", flags.count("0"))
subprocess.run(["rm", "-rf", tmp])
