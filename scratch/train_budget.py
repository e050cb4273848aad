"""Cohort training at the reference's real budget (scripts/opt_prior2.json:3-6: random_init_num 1000, top_iteration_num 40,
iteration_num_per_update 30) through medgp_train's continuous admission.  usage: python scratch/train_budget.py [P] [N] [resident] [extra trainer args...]"""
import os
import re
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medgp_amd.synth_experiment import make_experiment

P = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
RES = sys.argv[3] if len(sys.argv) > 3 else "1024"
host = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "medgp_amd", "host")
tmp = tempfile.mkdtemp(prefix="medgp_budget_")
pans = [f"P{k:05d}" for k in range(P)]
t0 = time.perf_counter()
ex = make_experiment(os.path.join(tmp, "train"), pans, D=24, Q=5, R=8, N=N, feature_index=tuple(range(24)), seed=77,
                     opt=dict(random_init_num=1000, top_iteration_num=40, iteration_num_per_update=30))
print(f"experiment written in {time.perf_counter() - t0:.1f} s", flush=True)
plist = os.path.join(tmp, "pans.txt")
open(plist, "w").write("\n".join(pans) + "\n")
t0 = time.perf_counter()
r = subprocess.run([os.path.join(host, "medgp_train"), "--cfg", ex["cfg"], "--pan-list", plist, "--resident", RES] + sys.argv[4:],
                   capture_output=True, text=True, timeout=3000)
wall = time.perf_counter() - t0
print("rc", r.returncode, "process wall", round(wall, 2), "s")
for ln in r.stdout.splitlines():
    if ln.startswith(("INFO: lock-step", "INFO: continuous", "INFO: gradient", "optimization finished", "ERROR", "Finish all")) or "merged" in ln:
        print(ln)
subprocess.run(["rm", "-rf", tmp])
