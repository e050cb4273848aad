"""rocprofv3 kernel trace of medgp_train on the heavy-tailed cohort at the real budget (python only writes the experiment)"""
import csv, glob, os, subprocess, sys, tempfile, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medgp_amd import synth
from medgp_amd.synth_experiment import make_experiment
P = 512
out = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_train_ragged")
host = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "medgp_amd", "host")
tmp = tempfile.mkdtemp(prefix="medgp_rg_")
ns = [max(48, int(v)) for v in synth.ragged_sizes(0, P)]
pans = [f"P{k:05d}" for k in range(P)]
ex = make_experiment(os.path.join(tmp, "train"), pans, D=24, Q=5, R=8, N=ns, feature_index=tuple(range(24)), seed=78,
                     opt=dict(random_init_num=1000, top_iteration_num=40, iteration_num_per_update=30))
plist = os.path.join(tmp, "pans.txt"); open(plist, "w").write("\n".join(pans) + "\n")
r = subprocess.run(["rocprofv3", "--kernel-trace"] + (["--hip-trace", "--stats"] if os.environ.get("HIPTRACE") else []) + ["--output-format", "csv", "-d", out, "--", os.path.join(host, "medgp_train"), "--cfg", ex["cfg"], "--pan-list", plist, "--resident", "512"],
                   capture_output=True, text=True, timeout=3000, env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp")
for ln in r.stdout.splitlines():
    if ln.startswith(("INFO: lock-step", "INFO: continuous", "optimization finished", "ERROR")): print(ln)
tr = glob.glob(os.path.join(out, "**", "*_kernel_trace.csv"), recursive=True)[0]
rows = sorted((int(x["Start_Timestamp"]), int(x["End_Timestamp"]), x["Kernel_Name"].split("(")[0].replace("void ", "")) for x in csv.DictReader(open(tr)))
# the screening phase = before the first k_wgrad
t_w = next(a for a, b, k in rows if k.startswith("k_wgrad"))
for name, sel in (("screening (before the first k_wgrad)", [x for x in rows if x[0] < t_w]), ("lock-step", [x for x in rows if x[0] >= t_w])):
    per = collections.defaultdict(lambda: [0, 0])
    for a, b, k in sel: per[k][0] += b - a; per[k][1] += 1
    span = sel[-1][1] - sel[0][0]
    print(f"{name}: span {span / 1e9:.3f} s")
    for k, (t, n) in sorted(per.items(), key=lambda kv: -kv[1][0])[:7]: print(f"   {k[:40]:40s} {t / 1e9:7.3f} s {n:7d} launches")
subprocess.run(["rm", "-rf", tmp])
for f in glob.glob(os.path.join(out, "**", "*hip_api_stats.csv"), recursive=True) + glob.glob(os.path.join(out, "**", "*hip_stats.csv"), recursive=True):
    print("---", os.path.basename(f))
    print("".join(open(f).readlines()[:12]))
