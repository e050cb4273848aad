"""BASELINE config 1 END TO END on the REFERENCE-WRITTEN configuration (round 6):
"2-output PT/INR (feature_PT_INR.json), 1 patient N ~ 150" -- D = 2, Q = 5, R = 2, H = 42, prior mode 2, at the reference's full
budget of scripts/opt_prior2.json: 1000 random initialisations, 40 variational-EM iterations (5 x 100 + 35 x 30 evaluations).

tests/golden/ref_cfg/PT_INR/exp_setup.json and hyp_bound.txt were written by the reference's own config.py, kernel/fold0/gmm_mode_* by
its binaryIO.py (tests/golden/make_golden.py::ref_config_files); they are used here BYTE FOR BYTE: the tree is copied under a scratch
directory that keeps the relative paths the file names (tests/golden/ref_cfg/PT_INR/...), one synthetic patient PT0001 with 75 + 75
observations is written next to it as feature18.txt / feature19.txt + feature<idx>_stat.bin in the reference's text format, and the
hosts run from that directory exactly as the reference's CLI is run from its repository root:
    medgp_train --cfg tests/golden/ref_cfg/PT_INR/exp_setup.json --pan PT0001 --thread 1        (ref: main_one_train.cpp:154-324)
    medgp_test  --cfg ... --pan PT0001 --thread 1 --fold 0 --kernclust-alg gmm                   (ref: main_one_test.cpp:190-480)
Expected values: the oracle optimiser (oracle/optimizer_oracle.py) and the restated imputation loop, both driven by the CPU oracle.
"""
import os
import subprocess
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from medgp_amd import synth
from oracle import optimizer_oracle as OO
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "medgp_amd", "host")
from medgp_amd.synth_experiment import CONFIG1_REL as REL, reference_config1_tree  # noqa: E402

PAN, Q, D, R, H = "PT0001", 5, 2, 2, 42
FEATS = (18, 19)


def make_tree(tmp):
    cwd, cfg, m, t, y = reference_config1_tree(tmp, ROOT, PAN)
    assert (m == 0).sum() == 75 and (m == 1).sum() == 75
    return cwd, cfg, m, t, y


def test_config1_train_and_test_on_the_reference_written_config(tmp_path, built_lib):
    for exe in ("medgp_train", "medgp_test", "host_logic_test"):
        if not os.path.exists(os.path.join(HOST, exe)):
            subprocess.check_call(["make", "-s", "-C", HOST, exe])
    cwd, cfg, m, t, y = make_tree(tmp_path)
    assert open(os.path.join(cwd, cfg), "rb").read() == open(os.path.join(ROOT, cfg), "rb").read()    # the reference's bytes, untouched
    # ---------------- medgp_train at the reference's budget
    t0 = time.perf_counter()
    r = subprocess.run([os.path.join(HOST, "medgp_train"), "--cfg", cfg, "--pan", PAN, "--thread", "1"], capture_output=True, text=True, timeout=600, cwd=cwd)
    wall_train = time.perf_counter() - t0
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    train = os.path.join(cwd, REL, "train")
    assert sorted(os.listdir(train)) == sorted(f"train_{k}_{PAN}.{e}" for k, e in (("init_hyp", "bin"), ("hyp", "bin"), ("var_hyp", "bin"), ("num", "txt"), ("flag", "txt")))
    assert open(os.path.join(train, f"train_num_{PAN}.txt")).read() == "150\n" and open(os.path.join(train, f"train_flag_{PAN}.txt")).read() == "1\n"
    # what the host loaded is what was written (its own dump)
    db = os.path.join(cwd, "data.bin")
    subprocess.check_call([os.path.join(HOST, "host_logic_test"), "data", cfg, PAN, db], stdout=subprocess.DEVNULL, cwd=cwd)
    raw = open(db, "rb").read()
    n = int(np.frombuffer(raw, np.int32, 1)[0])
    assert n == 150
    assert np.array_equal(np.frombuffer(raw, np.int32, n, 4), m) and np.array_equal(np.frombuffer(raw, np.float32, n, 4 + 4 * n), t)
    assert np.array_equal(np.frombuffer(raw, np.float32, n, 4 + 8 * n), y)
    # ---- f2: the 1000 initial vectors (glibc rand() through the reference-written bounds) and the screening arg-min
    hb = os.path.join(cwd, "hyp.bin")
    subprocess.check_call([os.path.join(HOST, "host_logic_test"), "hyp", cfg, hb], stdout=subprocess.DEVNULL, cwd=cwd)
    inits = np.fromfile(hb, np.float64).reshape(1000, H)
    best, best_init = np.inf, None
    for th in inits:
        rr = O.nlml_grad(7, Q, D, R, m, t, y, th, flag_grad=False)
        assert rr["ok"]
        if rr["nlml"] < best:
            best, best_init = rr["nlml"], th
    got_init = np.fromfile(os.path.join(train, f"train_init_hyp_{PAN}.bin"), np.float64)
    assert np.array_equal(got_init, best_init)
    # ---- f1: variational EM, 40 outer iterations (5 x 100 + 35 x 30 evaluations, early stop at < 0.5 % loss change), oracle objective
    nev = [0]

    def objective(pr):
        def obj(th):
            nev[0] += 1
            rr = O.nlml_grad(7, Q, D, R, m, t, y, np.asarray(th, np.float64), flag_grad=True, prior=pr)
            return (True, rr["nlml"], list(rr["grad"])) if rr["ok"] else (False, 0.0, [])
        return obj

    vp = OO.VarEMPrior(Q, D, R, 0.01)

    def obj_of_prior(v):
        pr = O.Prior.hier_gamma(Q, D, R, 0.01, 0.01)
        pr.type[D:D + Q * D * R] = np.array(v.type_A, np.int32)
        pr.p1[D:D + Q * D * R] = np.array(v.var_A, np.float32)
        return objective(pr)

    loss, theta, trace = OO.varem(-40, best_init, obj_of_prior, vp, D, 30)
    got = np.fromfile(os.path.join(train, f"train_hyp_{PAN}.bin"), np.float64)
    var = np.fromfile(os.path.join(train, f"train_var_hyp_{PAN}.bin"), np.float64)
    assert got.size == H and var.size == 2 * Q * (D * R + R)
    import re
    it_loss = [float(v) for v in re.findall(r"iteration \d+ for variational EM: loss = ([0-9.eE+-]+)", r.stdout)]
    gpu_loss = float(re.search(r"final loss = ([0-9.eE+-]+)", r.stdout).group(1))
    evals = int(re.search(r"optimization finished: (\d+) nlml\+grad evaluations", r.stdout).group(1))
    early = "meets early stop criterion" in r.stdout
    print(f"config 1: medgp_train wall {wall_train:.2f} s; outer iterations oracle {len(trace)} / device {len(it_loss)}, evaluations oracle {nev[0]} / device {evals}; "
          f"final loss oracle {loss:.9g} / device {gpu_loss:.9g}; max |dtheta| {np.abs(got - np.array(theta)).max():.3e}")
    print("   per-iteration loss, oracle vs device (6 digits printed):", [(round(a[0], 3), b) for a, b in zip(trace[:8], it_loss[:8])])
    # The two runs are the same algorithm on objectives that agree to ~1e-14, but 40 outer iterations of a sparse-prior optimisation
    # are not comparable point by point: weakly determined A entries amplify last-bit differences (1e-2 after 300 evaluations at
    # Q = 3, tests/test_train_host_gpu.py), line searches then take different decisions and the early-stop test (< 0.5 % loss change,
    # ref: c_optimizer_varEM.cpp:89-95) fires at different iterations.  What IS comparable:
    #  (1) the first outer iteration's loss to the 6 significant digits the host prints (as the reference does), the second -- 200
    #      evaluations in -- to 1e-4 (observed 2e-5: the divergence is already under way);
    assert len(it_loss) >= 3 and len(trace) >= 3
    assert abs(it_loss[0] - trace[0][0]) <= 3e-6 * abs(trace[0][0]), (it_loss[0], trace[0][0])
    assert abs(it_loss[1] - trace[1][0]) <= 1e-4 * abs(trace[1][0]), (it_loss[1], trace[1][0])
    #  (2) both end at a comparable optimum: final losses within 0.5 % (the optimiser's own stopping resolution), below the start;
    assert abs(gpu_loss - loss) <= 5e-3 * abs(loss) and gpu_loss < it_loss[0]
    #  (3) the END STATE is self-consistent under the oracle: with an early stop the variational state on file is the one the last
    #      sub-optimisation ran with, and the oracle's objective at the device's trained hypers under that prior is the device's final loss
    if early:
        psi = var[:Q * D * R]
        pr = O.Prior.hier_gamma(Q, D, R, 0.01, 0.01)
        pr.p1[D:D + Q * D * R] = psi.astype(np.float32)
        pr.type[D:D + Q * D * R] = np.where(psi == 0.0, 0, 1)
        rr = O.nlml_grad(7, Q, D, R, m, t, y, got, flag_grad=False, prior=pr)
        assert rr["ok"] and abs(rr["nlml"] - gpu_loss) <= 3e-6 * abs(gpu_loss), (rr["nlml"], gpu_loss)
    #  (4) ONE outer iteration point by point: the same reference-written file with top_iteration_num set to 1 (the only edit)
    cfg1 = os.path.join(REL, "exp_setup_1iter.json")
    txt = open(os.path.join(cwd, cfg)).read()
    assert '"top_iteration_num": 40' in txt
    open(os.path.join(cwd, cfg1), "w").write(txt.replace('"top_iteration_num": 40', '"top_iteration_num": 1'))
    for f in os.listdir(train):
        os.remove(os.path.join(train, f))
    r1 = subprocess.run([os.path.join(HOST, "medgp_train"), "--cfg", cfg1, "--pan", PAN, "--thread", "1"], capture_output=True, text=True, timeout=600, cwd=cwd)
    assert r1.returncode == 0, r1.stdout[-3000:] + r1.stderr[-2000:]
    vp1 = OO.VarEMPrior(Q, D, R, 0.01)
    loss1, theta1, _ = OO.varem(-1, best_init, obj_of_prior, vp1, D, 30)
    got1 = np.fromfile(os.path.join(train, f"train_hyp_{PAN}.bin"), np.float64)
    var1 = np.fromfile(os.path.join(train, f"train_var_hyp_{PAN}.bin"), np.float64)
    err1 = np.abs(got1 - np.array(theta1)) / np.maximum(1.0, np.abs(theta1))
    print(f"   one outer iteration (100 evaluations): max theta err {err1.max():.3e}, var-EM state max abs diff {np.abs(var1 - np.array(vp1.cov_varEM)).max():.3e}")
    assert err1.max() <= 2e-6      # (observed 4.7e-7: 100 evaluations of a 42-hyper sparse-prior fit amplify the objectives' 1e-14 that far)
    np.testing.assert_allclose(var1, np.array(vp1.cov_varEM), rtol=1e-4, atol=1e-6)
    # ---------------- medgp_test, both passes, on the reference-written mode kernel (Q = 3 after clustering)
    t0 = time.perf_counter()
    r = subprocess.run([os.path.join(HOST, "medgp_test"), "--cfg", cfg, "--pan", PAN, "--thread", "1", "--fold", "0", "--kernclust-alg", "gmm"],
                       capture_output=True, text=True, timeout=600, cwd=cwd)
    wall_test = time.perf_counter() - t0
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    from test_test_host_gpu import reference_loop
    mode = np.load(os.path.join(ROOT, REL, "mode_expected.npy"))
    Qm = int(open(os.path.join(cwd, REL, "kernel", "fold0", "gmm_mode_mixture_num.txt")).read())
    assert Qm == 3 and mode.size == D + Qm * (D * R + 2 + D)
    cfgj = __import__("json").load(open(os.path.join(cwd, cfg)))
    lr, mom = cfgj["online_learn_rate"], cfgj["online_momentum"]
    test = os.path.join(cwd, REL, "test")
    for flag_update, name in ((False, "mean_wo_update"), (True, "mean_w_update")):
        feat, ci, et, err, pred = reference_loop(m, t, y, mode, Qm, D, R, flag_update, lr, mom, list(FEATS))
        pre = os.path.join(test, f"test_{name}_")
        assert open(pre + f"flag_{PAN}.txt").read() == "1\n"
        assert [int(v) for v in open(pre + f"feature_{PAN}.txt").read().split()] == feat
        np.testing.assert_array_equal(np.fromfile(pre + f"etime_{PAN}.bin", np.float64), et)
        np.testing.assert_allclose(np.fromfile(pre + f"pred_{PAN}.bin", np.float64), pred, rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(np.fromfile(pre + f"error_{PAN}.bin", np.float64), err, rtol=2e-5, atol=2e-6)
        gc = [int(v) for v in open(pre + f"ci_{PAN}.txt").read().split()]
        assert sum(a != b for a, b in zip(gc, ci)) <= 1
    print(f"config 1: medgp_test wall {wall_test:.2f} s (both passes, 150 imputations each)")
