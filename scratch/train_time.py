"""end-to-end cohort training time: builds a synthetic experiment (P patients, D=24, N=512, Q=5, R=8, prior mode 2) in the
reference's file formats and runs medgp_train --pan-list on it.  usage: python scratch/train_time.py [P] [N] [extra medgp_train arguments ...]"""
import os, sys, subprocess, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from exp_fixture import make_experiment
P = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
D = 24
tmp = tempfile.mkdtemp(prefix="medgp_train_time_")
pans = [f"P{k:04d}" for k in range(P)]
t0 = time.time()
ex = make_experiment(tmp, pans, D=D, Q=5, R=8, N=N, feature_index=tuple(range(D)), opt=dict(random_init_num=20, top_iteration_num=1))
plist = os.path.join(tmp, "pans.txt"); open(plist, "w").write("\n".join(pans) + "\n")
print(f"experiment written in {time.time() - t0:.1f} s")
t0 = time.time()
out = subprocess.run([os.path.join(ROOT, "medgp_amd", "host", "medgp_train"), "--cfg", ex["cfg"], "--pan-list", plist, "--thread", "1"] + sys.argv[3:],
                     capture_output=True, text=True)
print(f"medgp_train wall {time.time() - t0:.2f} s, rc {out.returncode}")
for l in out.stdout.splitlines():
    if l.startswith("INFO: lock") or l.startswith("optimization finished") or l.startswith("Finish all") or "ERROR" in l or l.startswith("INFO: init") or l.startswith("INFO: loaded"):
        print("  ", l)
print(out.stderr[-500:])
