"""the short-budget host leg of bench.py (512 patients, 20 initial vectors, one varEM iteration): where does the trainer's wall go?"""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medgp_amd.synth_experiment import make_experiment
host = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "medgp_amd", "host")
tmp = tempfile.mkdtemp(prefix="medgp_short_")
pans = [f"P{k:04d}" for k in range(512)]
ex = make_experiment(os.path.join(tmp, "train"), pans, D=24, Q=5, R=8, N=512, feature_index=tuple(range(24)), seed=2027, opt=dict(random_init_num=20, top_iteration_num=1))
plist = os.path.join(tmp, "p.txt"); open(plist, "w").write("\n".join(pans) + "\n")
for rep in range(3):
    t0 = time.perf_counter()
    r = subprocess.run([os.path.join(host, "medgp_train"), "--cfg", ex["cfg"], "--pan-list", plist] + sys.argv[1:], capture_output=True, text=True, timeout=600)
    print("rc", r.returncode, "wall", round(time.perf_counter() - t0, 3))
    for ln in r.stdout.splitlines():
        if ln.startswith(("INFO: lock-step", "INFO: continuous", "optimization finished")): print("  ", ln)
subprocess.run(["rm", "-rf", tmp])
