"""GPU test of medgp_test (the main_one_test replacement): both passes of the online imputation loop against a
Python restatement of ref main_one_test.cpp:269-444 driven by the oracle."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from exp_fixture import make_experiment
from medgp_amd import synth
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "medgp_amd", "host")
EXE = os.path.join(HOST, "medgp_test")


def loaded(ex, pan, D):
    m, t, y = [], [], []
    for j in range(D):
        tt = np.array([np.float32(f"{a:.6f}") for a in ex["raw"][pan][j][0]], np.float32)
        vv = np.array([np.float32(f"{a:.6f}") for a in ex["raw"][pan][j][1]], np.float32)
        m += [j] * len(tt)
        t.append(tt)
        y.append(((vv.astype(np.float64) - ex["stats"][j][0]) / ex["stats"][j][1]).astype(np.float32))
    return np.array(m, np.int32), np.concatenate(t), np.concatenate(y)


def reference_loop(m, t, y, mode, Q, D, R, flag_update, lr, mom, feature_index):
    return reference_loop_generic(7, Q, D, R, m, t, y, mode, flag_update, lr, mom, feature_index)


def reference_loop_generic(kidx, Q, D, R, m, t, y, mode, flag_update, lr, mom, feature_index):
    """ref main_one_test.cpp:269-444 restated with the oracle as the GP (kidx 7: LMC-SM with the test-time clamp of the
    exactly-zero A entries, ref c_prior.cpp:118-140; kidx 0 / 8: the single-output families, no test-time prior, m = None)."""
    uniq = np.unique(t)
    best, delta = mode.copy(), np.zeros_like(mode)
    H = mode.size
    clamp = np.zeros(H, bool)
    if kidx == 7:
        clamp[D:D + Q * D * R] = mode[D:D + Q * D * R] == 0.0
    pr = O.Prior(H)
    pr.flag[clamp] = 1
    pr.type[clamp] = 0
    mm = np.zeros(t.size, np.int32) if m is None else m
    sub = (lambda idx: None) if m is None else (lambda idx: m[idx])
    last = uniq[0]
    feat, ci, et, err, pred = [], [], [], [], []
    for tt, tu in enumerate(uniq):
        past = [i for i in range(t.size) if t[i] < tu and (not flag_update or abs(np.float32(t[i] - tu)) <= 72.0)]
        curr = [i for i in range(t.size) if t[i] == tu]
        if flag_update and tt > 3 and np.float32(tu - last) > 5.0 / 60.0:
            last = tu
            r = O.nlml_grad(kidx, Q, D, R, sub(past), t[past], y[past], best, prior=pr) if len(past) > 2 else {"ok": False}
            if r["ok"]:
                upd = ~clamp
                delta[upd] = mom * delta[upd] + lr * r["grad"][upd]
                best[upd] -= delta[upd]
            else:
                best, delta = mode.copy(), np.zeros_like(mode)
        for jj, it in enumerate(curr):
            tr = past + [c for k, c in enumerate(curr) if k != jj]
            if tr:
                rp = O.fit_predict(kidx, Q, D, R, sub(tr), t[tr], y[tr], best, sub([it]), t[[it]])
                mu, var = np.float32(rp["mean"][0]), np.float32(rp["var"][0])
                e = float(np.float32(mu - y[it]))
                pred.append(float(mu)); err.append(e); ci.append(int(abs(e) <= 1.96 * np.sqrt(var)))
            else:
                e = float(np.float32(0.0 - float(y[it])))
                pred.append(0.0); err.append(0.0 - float(y[it])); ci.append(int(abs(e) <= 1.96 * np.exp(mode[mm[it]])))
            feat.append(feature_index[mm[it]]); et.append(float(np.float32(t[it] - tu)))
    return feat, ci, et, err, pred


def test_online_imputation_matches_oracle_loop(tmp_path, built_lib):
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-s", "-C", HOST, "medgp_test"])
    Q, D, R, N = 3, 2, 2, 46
    ex = make_experiment(tmp_path, ["P007"], D=D, Q=Q, R=R, N=N, opt={"online_learn_rate": 1e-4})
    mode = synth.theta(9, 0, 7, Q, D, R)
    mode[D + 1] = 0.0
    mode[D + 5] = 0.0          # exactly-zero A entries -> clamped by init_test_prior
    fold_dir = os.path.join(ex["dirs"]["kernel"], "fold0")
    os.makedirs(fold_dir)
    open(os.path.join(fold_dir, "gmm_mode_mixture_num.txt"), "w").write(f"{Q}\n")
    mode.tofile(os.path.join(fold_dir, "gmm_mode_param.bin"))
    r = subprocess.run([EXE, "--cfg", ex["cfg"], "--pan", "P007", "--thread", "1", "--fold", "0", "--kernclust-alg", "gmm",
                        "--max-batch", "16"], capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    m, t, y = loaded(ex, "P007", D)
    for flag_update, mode_name in ((False, "mean_wo_update"), (True, "mean_w_update")):
        feat, ci, et, err, pred = reference_loop(m, t, y, mode, Q, D, R, flag_update, 1e-4, 0.9, ex["feature_index"])
        pre = os.path.join(ex["dirs"]["test"], f"test_{mode_name}_")
        assert open(pre + "flag_P007.txt").read() == "1\n"
        gf = [int(v) for v in open(pre + "feature_P007.txt").read().split()]
        gc = [int(v) for v in open(pre + "ci_P007.txt").read().split()]
        ge = np.fromfile(pre + "error_P007.bin", np.float64)
        gp = np.fromfile(pre + "pred_P007.bin", np.float64)
        gt = np.fromfile(pre + "etime_P007.bin", np.float64)
        assert len(gf) == t.size and gf == feat
        np.testing.assert_array_equal(gt, et)
        np.testing.assert_allclose(gp, pred, rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(ge, err, rtol=2e-5, atol=2e-6)
        # the CI flag may legitimately differ only where |error| sits within float rounding of the bound
        diff = [k for k in range(len(gc)) if gc[k] != ci[k]]
        assert len(diff) <= 1, diff


def test_cohort_list_is_byte_identical_to_one_run_per_patient(tmp_path, built_lib):
    """medgp_test --pan-list (hyper trajectories of all patients in lock step, one shared factorisation call, all imputation
    problems of the cohort packed into batches) writes exactly the bytes that one `medgp_test --pan` run per patient writes:
    three patients of different sizes, one of them with too few samples for any update (ref: main_one_test.cpp:308, the
    n > 2 guard of util/c_objective_one.cpp:51) and a different batch cap, in both passes."""
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-s", "-C", HOST, "medgp_test"])
    Q, D, R = 3, 2, 2
    pans = ["P101", "P102", "P103", "P104"]      # P102: too few samples for any update; P104: no samples at all (flag file 0)
    ex = make_experiment(tmp_path, pans, D=D, Q=Q, R=R, N=[46, 5, 70, 0], opt={"online_learn_rate": 1e-4})
    mode = synth.theta(9, 0, 7, Q, D, R)
    mode[D + 1] = 0.0
    fold_dir = os.path.join(ex["dirs"]["kernel"], "fold0")
    os.makedirs(fold_dir)
    open(os.path.join(fold_dir, "gmm_mode_mixture_num.txt"), "w").write(f"{Q}\n")
    mode.tofile(os.path.join(fold_dir, "gmm_mode_param.bin"))
    plist = tmp_path / "pans.txt"
    plist.write_text("\n".join(pans) + "\n")
    base = [EXE, "--cfg", ex["cfg"], "--thread", "1", "--fold", "0", "--kernclust-alg", "gmm"]
    r = subprocess.run(base + ["--pan-list", str(plist), "--max-batch", "64"], capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "lock-step rounds" in r.stdout and "4 patient(s)" in r.stdout
    tdir = ex["dirs"]["test"]
    cohort = {f: open(os.path.join(tdir, f), "rb").read() for f in sorted(os.listdir(tdir)) if f.startswith("test_")}
    assert len(cohort) == 2 * (6 + 6 + 6 + 1)       # three patients x two passes x (feature, etime, ci, error, pred, flag) + the empty one's flags
    assert cohort["test_mean_wo_update_flag_P104.txt"] == b"0\n" and cohort["test_mean_w_update_flag_P101.txt"] == b"1\n"
    for f in cohort:
        os.remove(os.path.join(tdir, f))
    for pan in pans:
        r1 = subprocess.run(base + ["--pan", pan, "--max-batch", "16"], capture_output=True, text=True, timeout=240)
        assert r1.returncode == 0, r1.stdout[-3000:] + r1.stderr[-2000:]
    single = {f: open(os.path.join(tdir, f), "rb").read() for f in sorted(os.listdir(tdir)) if f.startswith("test_")}
    assert sorted(single) == sorted(cohort)
    for f in cohort:
        assert cohort[f] == single[f], f
    # a patient of the list whose files cannot be read: reported, skipped, the others' files are the same bytes, exit code non-zero
    # (the reference runs one process per patient, so one broken patient costs only its own outputs; advisor finding, round 4)
    for f in cohort:
        os.remove(os.path.join(tdir, f))
    plist2 = tmp_path / "pans_missing.txt"
    plist2.write_text("\n".join(pans[:2] + ["NOSUCH"] + pans[2:]) + "\n")
    r2 = subprocess.run(base + ["--pan-list", str(plist2), "--max-batch", "64"], capture_output=True, text=True, timeout=240)
    assert r2.returncode != 0 and "NOSUCH" in r2.stdout and "1 patient(s) could not be read" in r2.stdout
    again = {f: open(os.path.join(tdir, f), "rb").read() for f in sorted(os.listdir(tdir)) if f.startswith("test_")}
    assert again == cohort
    # and the cohort's values are the oracle loop's (the single-patient test above pins the arithmetic; here one more patient)
    m, t, y = loaded(ex, "P103", D)
    feat, ci, et, err, pred = reference_loop(m, t, y, mode, Q, D, R, True, 1e-4, 0.9, ex["feature_index"])
    gp = np.frombuffer(cohort["test_mean_w_update_pred_P103.bin"], np.float64)
    np.testing.assert_allclose(gp, pred, rtol=2e-5, atol=2e-6)
