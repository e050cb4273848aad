"""Cohort form of medgp_test vs one process per patient (what the reference's scheduler fan-out does, ref: scripts/test_della.sh:46):
wall time of both passes for a synthetic cohort of P patients x N observations (D = 4), and byte-identity of the outputs.
usage: python scratch/impute_cohort_time.py [P=64] [N=200] [nseq=16]   (nseq sequential single-patient runs are timed and scaled to P)"""
import os, subprocess, sys, tempfile, time, re
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from exp_fixture import make_experiment
from medgp_amd import synth
EXE = os.path.join(ROOT, "medgp_amd", "host", "medgp_test")
P = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
nseq = int(sys.argv[3]) if len(sys.argv) > 3 else 16
Q, D, R = 3, 4, 2
pans = [f"C{k:03d}" for k in range(P)]
rng = np.random.default_rng(1)
Ns = [int(v) for v in rng.integers(int(0.6 * N), N + 1, size=P)]
td = tempfile.mkdtemp()
ex = make_experiment(td, pans, D=D, Q=Q, R=R, N=Ns, feature_index=(18, 19, 20, 21), opt={"online_learn_rate": 1e-4})
th = synth.theta(9, 0, 7, Q, D, R)
fold_dir = os.path.join(ex["dirs"]["kernel"], "fold0"); os.makedirs(fold_dir)
open(os.path.join(fold_dir, "gmm_mode_mixture_num.txt"), "w").write(f"{Q}\n"); th.tofile(os.path.join(fold_dir, "gmm_mode_param.bin"))
plist = os.path.join(td, "pans.txt"); open(plist, "w").write("\n".join(pans) + "\n")
base = [EXE, "--cfg", ex["cfg"], "--thread", "1", "--fold", "0", "--kernclust-alg", "gmm"]
tdir = ex["dirs"]["test"]
def snapshot():
    out = {f: open(os.path.join(tdir, f), "rb").read() for f in sorted(os.listdir(tdir)) if f.startswith("test_")}
    for f in out: os.remove(os.path.join(tdir, f))
    return out
for extra, tag in (([], "pinned route (default)"), (["--auto-route"], "--auto-route")):
    t0 = time.perf_counter(); r = subprocess.run(base + ["--pan-list", plist] + extra, capture_output=True, text=True); dt = time.perf_counter() - t0
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    print(f"cohort of {P} patients (N {min(Ns)}..{max(Ns)}), {tag}: process wall {dt:.2f} s")
    for l in r.stdout.split("\n"):
        if l.startswith("INFO:"): print("   ", l)
    if not extra: cohort = snapshot(); t_cohort = dt
    else: snapshot()
t0 = time.perf_counter()
for pan in pans[:nseq]:
    r = subprocess.run(base + ["--pan", pan], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
t_seq = time.perf_counter() - t0
single = snapshot()
same = all(cohort[f] == single[f] for f in single)
print(f"{nseq} sequential single-patient runs: {t_seq:.2f} s  -> {P} runs ~ {t_seq * P / nseq:.1f} s;  cohort run {t_cohort:.2f} s = 1/{t_seq * P / nseq / t_cohort:.1f};  outputs of those {nseq} patients byte-identical: {same} ({len(single)} files)")
pw = [float(x) for x in re.findall(r"pass wall time: without updating ([0-9.e+-]+) ms, with updating ([0-9.e+-]+) ms", r.stdout)[0]]
print(f"last single run passes: without updating {pw[0]:.1f} ms, with updating {pw[1]:.1f} ms")
