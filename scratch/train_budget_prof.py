"""rocprofv3 kernel trace of medgp_train at the reference's real budget: how busy is the device?  (python writes the experiment and
never touches the GPU; the trainer itself is the program after `--`.)  usage: python scratch/train_budget_prof.py [P] [out_dir]"""
import csv, glob, os, subprocess, sys, tempfile, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medgp_amd.synth_experiment import make_experiment
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
out = os.path.abspath(sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/prof_train")
host = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "medgp_amd", "host")
tmp = tempfile.mkdtemp(prefix="medgp_budget_")
pans = [f"P{k:05d}" for k in range(P)]
ex = make_experiment(os.path.join(tmp, "train"), pans, D=24, Q=5, R=8, N=512, feature_index=tuple(range(24)), seed=77,
                     opt=dict(random_init_num=1000, top_iteration_num=40, iteration_num_per_update=30))
plist = os.path.join(tmp, "pans.txt")
open(plist, "w").write("\n".join(pans) + "\n")
env = dict(os.environ, TMPDIR="/tmp")
t0 = time.perf_counter()
r = subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", out, "--",
                    os.path.join(host, "medgp_train"), "--cfg", ex["cfg"], "--pan-list", plist], capture_output=True, text=True, timeout=3000, env=env, cwd="/tmp")
print("rc", r.returncode, "wall", round(time.perf_counter() - t0, 2))
for ln in r.stdout.splitlines():
    if ln.startswith(("INFO: lock-step", "INFO: continuous", "INFO: gradient", "optimization finished", "ERROR")):
        print(ln)
tr = glob.glob(os.path.join(out, "**", "*_kernel_trace.csv"), recursive=True)
if tr:
    per = collections.defaultdict(lambda: [0, 0])
    tmin, tmax, busy = None, 0, 0
    iv = []
    for row in csv.DictReader(open(tr[0])):
        a, b = int(row["Start_Timestamp"]), int(row["End_Timestamp"])
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        per[k][0] += b - a; per[k][1] += 1
        iv.append((a, b))
    iv.sort()
    cur_a, cur_b = iv[0]
    for a, b in iv[1:]:
        if a <= cur_b: cur_b = max(cur_b, b)
        else: busy += cur_b - cur_a; cur_a, cur_b = a, b
    busy += cur_b - cur_a
    span = iv[-1][1] - iv[0][0]
    print(f"kernels: first start to last end {span / 1e9:.3f} s; union of kernel intervals {busy / 1e9:.3f} s = {busy / span:.3f} of it")
    for k, (ns, n) in sorted(per.items(), key=lambda kv: -kv[1][0])[:10]:
        print(f"  {k[:40]:40s} {ns / 1e9:8.3f} s  {n:7d} launches")
subprocess.run(["rm", "-rf", tmp])
