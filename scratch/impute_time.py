"""Wall time of medgp_test's imputation passes on one synthetic 300-observation patient (D = 4): shared factorisation vs
one factorisation per imputed observation (--per-problem), and agreement of the two outputs."""
import os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from exp_fixture import make_experiment
from medgp_amd import synth
EXE = os.path.join(ROOT, "medgp_amd", "host", "medgp_test")
Q, D, R, N = 3, 4, 2, int(sys.argv[1]) if len(sys.argv) > 1 else 300
res = {}
for mode in ("shared", "per-problem"):
    td = tempfile.mkdtemp()
    ex = make_experiment(td, ["P300"], D=D, Q=Q, R=R, N=N, feature_index=(18, 19, 20, 21), opt={"online_learn_rate": 1e-4})
    th = synth.theta(9, 0, 7, Q, D, R)
    fold_dir = os.path.join(ex["dirs"]["kernel"], "fold0"); os.makedirs(fold_dir)
    open(os.path.join(fold_dir, "gmm_mode_mixture_num.txt"), "w").write(f"{Q}\n"); th.tofile(os.path.join(fold_dir, "gmm_mode_param.bin"))
    args = [EXE, "--cfg", ex["cfg"], "--pan", "P300", "--thread", "1", "--fold", "0", "--kernclust-alg", "gmm"] + (["--per-problem"] if mode == "per-problem" else [])
    t0 = time.perf_counter(); r = subprocess.run(args, capture_output=True, text=True); dt = time.perf_counter() - t0
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    print(mode, f"process wall {dt:.2f} s;", " | ".join(l for l in r.stdout.split("\n") if l.startswith("INFO:")))
    res[mode] = {k: np.fromfile(os.path.join(ex["dirs"]["test"], f"test_{m}_{k}_P300.bin"), np.float64) for m in ("mean_wo_update",) for k in ("pred", "error")}
d = np.abs(res["shared"]["pred"] - res["per-problem"]["pred"])
print("pred[:6]", res["shared"]["pred"][:6], res["per-problem"]["pred"][:6])
print("no-update pass: max |pred(shared) - pred(per problem)| =", d.max(), "over", d.size, "imputations")
