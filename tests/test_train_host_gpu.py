"""GPU test of medgp_train (the main_one_train replacement): file surface, lock-step cohort training ==
one-patient-at-a-time training bit for bit, losses decrease, prior-mode-2 state is written."""
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from exp_fixture import make_experiment
from medgp_amd import synth
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "medgp_amd", "host")
EXE = os.path.join(HOST, "medgp_train")


def run(args, timeout=180):
    r = subprocess.run([EXE] + args, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    return r.stdout


@pytest.mark.parametrize("prior_index", [2, 0])
def test_train_single_vs_cohort(tmp_path, built_lib, prior_index):
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-s", "-C", HOST, "medgp_train"])
    pans = ["P001", "P002", "P003"]
    exA = make_experiment(tmp_path / "a", pans, D=2, Q=3, R=2, N=[60, 75, 48], prior_index=prior_index)
    exB = make_experiment(tmp_path / "b", pans, D=2, Q=3, R=2, N=[60, 75, 48], prior_index=prior_index)
    for pan in pans:                                   # reference CLI: one patient per process
        run(["--cfg", exA["cfg"], "--pan", pan, "--thread", "1"])
    plist = tmp_path / "pans.txt"
    plist.write_text("\n".join(pans) + "\n")
    out = run(["--cfg", exB["cfg"], "--pan-list", str(plist)])   # cohort mode: lock step on one GPU
    assert "lock-step batches" in out
    Q, D, R = 3, 2, 2
    H = D + Q * (D * R + 2 + D)
    for p, pan in enumerate(pans):
        fa, fb = exA["dirs"]["train"], exB["dirs"]["train"]
        for name, count in (("train_init_hyp_", H), ("train_hyp_", H)):
            a = np.fromfile(os.path.join(fa, name + pan + ".bin"), np.float64)
            b = np.fromfile(os.path.join(fb, name + pan + ".bin"), np.float64)
            assert a.size == count and np.array_equal(a, b), (name, pan)
        if prior_index == 2:
            v = np.fromfile(os.path.join(fa, "train_var_hyp_" + pan + ".bin"), np.float64)
            assert v.size == 2 * Q * (D * R + R) and np.array_equal(v, np.fromfile(os.path.join(fb, "train_var_hyp_" + pan + ".bin"), np.float64))
        else:
            assert not os.path.exists(os.path.join(fa, "train_var_hyp_" + pan + ".bin"))
        assert open(os.path.join(fa, "train_flag_" + pan + ".txt")).read() == "1\n"
        n = int(open(os.path.join(fa, "train_num_" + pan + ".txt")).read())
        m, t, y = synth.patient(5, p, D, [60, 75, 48][p])
        assert n == t.size
        # the optimiser must have improved on the best random initial point (evaluated with the oracle, no prior)
        init = np.fromfile(os.path.join(fa, "train_init_hyp_" + pan + ".bin"), np.float64)
        fin = np.fromfile(os.path.join(fa, "train_hyp_" + pan + ".bin"), np.float64)
        # data as the host loads it (6-decimal text round trip)
        tt = np.concatenate([np.array([np.float32(f"{a:.6f}") for a in exA["raw"][pan][j][0]], np.float32) for j in range(D)])
        yy = np.concatenate([((np.array([np.float32(f"{a:.6f}") for a in exA["raw"][pan][j][1]], np.float32).astype(np.float64)
                               - exA["stats"][j][0]) / exA["stats"][j][1]).astype(np.float32) for j in range(D)])
        f_init = O.nlml_grad(7, Q, D, R, m, tt, yy, init, flag_grad=False)["nlml"]
        if prior_index == 0:
            f_fin = O.nlml_grad(7, Q, D, R, m, tt, yy, fin, flag_grad=False)["nlml"]
            assert f_fin < f_init


def test_train_skips_patient_with_too_few_samples(tmp_path, built_lib):
    ex = make_experiment(tmp_path, ["P009"], D=2, Q=2, R=2, N=3)   # one output gets a single observation
    run(["--cfg", ex["cfg"], "--pan", "P009", "--thread", "1"])
    assert open(os.path.join(ex["dirs"]["train"], "train_flag_P009.txt")).read() == "0\n"   # ref main_one_train.cpp:185-201
    assert open(os.path.join(ex["dirs"]["train"], "train_num_P009.txt")).read() == "3\n"
    assert not os.path.exists(os.path.join(ex["dirs"]["train"], "train_hyp_P009.bin"))


@pytest.mark.parametrize("prior_index", [2, 0])
def test_train_vs_reference_optimiser_restatement(tmp_path, built_lib, prior_index):
    """Rows f1 / f2 against an INDEPENDENT oracle: the screening arg-min (main_one_train.cpp:228-253) and the optimiser run
    (:258-300; c_optimizer_scg / c_optimizer_varEM restated in oracle/optimizer_oracle.py) driven by the CPU oracle's
    nlml + gradient, compared with the files medgp_train writes.  Same initial points (the glibc rand() draws are dumped by
    host_logic_test), same evaluation budgets; the two objective implementations agree to ~1e-14, the line searches take
    the same decisions, and the trained hypers agree far below the 1e-6 the north star asks of the objective itself."""
    from oracle import optimizer_oracle as OO
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-s", "-C", HOST, "medgp_train"])
    logic = os.path.join(HOST, "host_logic_test")
    if not os.path.exists(logic):
        subprocess.check_call(["make", "-s", "-C", HOST, "host_logic_test"])
    pans = ["P001", "P002", "P003"]
    Ns = [60, 75, 48]
    Q, D, R = 3, 2, 2
    H = D + Q * (D * R + 2 + D)
    # variational EM: ONE outer iteration (100 evaluations + the closed-form update).  Longer runs are not comparable point
    # by point: weakly determined A entries are shrunk towards zero by the sparse prior, and 1e-14 differences between the two
    # objective implementations grow to 1e-2 in those entries over 300 evaluations (observed); the state machines themselves
    # are compared bit for bit over several outer iterations on analytic objectives (tests/test_optimizer_oracle.py).
    opt = {"top_iteration_num": 1} if prior_index == 2 else {"top_iteration_num": 40}   # prior 0: 40 SCG evaluations
    ex = make_experiment(tmp_path / "e", pans, D=D, Q=Q, R=R, N=Ns, prior_index=prior_index, opt=opt)
    for pan in pans:
        run(["--cfg", ex["cfg"], "--pan", pan, "--thread", "1"])
    hb = tmp_path / "hyp.bin"
    subprocess.check_call([logic, "hyp", ex["cfg"], str(hb)], stdout=subprocess.DEVNULL)
    inits = np.fromfile(hb, np.float64).reshape(ex["opt"]["random_init_num"], H)
    for pan in pans:
        db = tmp_path / f"data_{pan}.bin"
        subprocess.check_call([logic, "data", ex["cfg"], pan, str(db)], stdout=subprocess.DEVNULL)
        raw = open(db, "rb").read()
        n = int(np.frombuffer(raw, np.int32, 1)[0])
        m = np.frombuffer(raw, np.int32, n, 4).copy()
        t = np.frombuffer(raw, np.float32, n, 4 + 4 * n).copy()
        y = np.frombuffer(raw, np.float32, n, 4 + 8 * n).copy()
        # ---- f2: screening, nlml only, no prior yet (prior set up after it, main_one_train.cpp:222-226, :258-264)
        best, best_init = np.inf, None
        for th in inits:
            r = O.nlml_grad(7, Q, D, R, m, t, y, th, flag_grad=False)
            assert r["ok"]
            if r["nlml"] < best:
                best, best_init = r["nlml"], th
        got_init = np.fromfile(os.path.join(ex["dirs"]["train"], "train_init_hyp_" + pan + ".bin"), np.float64)
        assert np.array_equal(got_init, best_init), pan
        # ---- f1: the optimiser
        nev = [0]

        def objective(pr):
            def obj(th):
                nev[0] += 1
                r = O.nlml_grad(7, Q, D, R, m, t, y, np.asarray(th, np.float64), flag_grad=True, prior=pr)
                if not r["ok"]:
                    return False, 0.0, []
                return True, r["nlml"], list(r["grad"])
            return obj

        budget = -ex["opt"]["top_iteration_num"]
        if prior_index == 2:
            vp = OO.VarEMPrior(Q, D, R, 0.01)

            def obj_of_prior(v):
                pr = O.Prior.hier_gamma(Q, D, R, 0.01, 0.01)
                a0 = D
                pr.type[a0:a0 + Q * D * R] = np.array(v.type_A, np.int32)
                pr.p1[a0:a0 + Q * D * R] = np.array(v.var_A, np.float32)
                return objective(pr)
            loss, theta, _ = OO.varem(budget, best_init, obj_of_prior, vp, D, ex["opt"]["iteration_num_per_update"])
            var = np.fromfile(os.path.join(ex["dirs"]["train"], "train_var_hyp_" + pan + ".bin"), np.float64)
            print(pan, "varEM state max abs diff", np.abs(var - np.array(vp.cov_varEM)).max())
            np.testing.assert_allclose(var, np.array(vp.cov_varEM), rtol=1e-4, atol=1e-6)
        else:
            loss, theta, _ = OO.scg(budget, best_init, objective(None))
        got = np.fromfile(os.path.join(ex["dirs"]["train"], "train_hyp_" + pan + ".bin"), np.float64)
        err = np.abs(got - np.array(theta)) / np.maximum(1.0, np.abs(theta))
        print(pan, "theta max err", float(err.max()), "evaluations", nev[0])
        assert err.max() <= 1e-6, (pan, float(err.max()))


@pytest.mark.parametrize("prior_index", [2, 0])
def test_train_pingpong_groups_and_host_threads_change_nothing(tmp_path, built_lib, prior_index):
    """The lock-step loop may split the active set into alternating groups (two asynchronous lanes: the host threads run one
    group's state machines while the device evaluates the other) and run the state machines on any number of host threads:
    every patient's files are byte-identical to the one-group, one-thread run (patients are independent; ragged sizes, one
    patient that finishes early through its evaluation budget)."""
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-s", "-C", HOST, "medgp_train"])
    pans = [f"P{k:03d}" for k in range(7)]
    Ns = [60, 75, 48, 52, 64, 33, 70]
    exs = []
    for tag, extra in (("a", ["--host-threads", "1", "--pingpong-min", "1000000"]), ("b", ["--host-threads", "4", "--pingpong-min", "2"]),
                       ("c", ["--host-threads", "3", "--pingpong-min", "1000000", "--max-batch", "3"])):
        ex = make_experiment(tmp_path / tag, pans, D=2, Q=3, R=2, N=Ns, prior_index=prior_index)
        plist = tmp_path / f"pans_{tag}.txt"
        plist.write_text("\n".join(pans) + "\n")
        out = run(["--cfg", ex["cfg"], "--pan-list", str(plist)] + extra)
        assert "lock-step batches" in out
        exs.append((ex, out))
    assert "1 group(s)" in exs[0][1] and "2 group(s)" in exs[1][1] and "3 group(s)" in exs[2][1]
    for pan in pans:
        for name in ["train_init_hyp_", "train_hyp_"] + (["train_var_hyp_"] if prior_index == 2 else []):
            a = np.fromfile(os.path.join(exs[0][0]["dirs"]["train"], name + pan + ".bin"), np.float64)
            for ex, _ in exs[1:]:
                b = np.fromfile(os.path.join(ex["dirs"]["train"], name + pan + ".bin"), np.float64)
                assert a.size > 0 and np.array_equal(a, b), (name, pan)


@pytest.mark.parametrize("prior_index", [2, 0])
def test_train_continuous_admission_matches_single_runs(tmp_path, built_lib, prior_index):
    """Round 5: the trainer keeps --resident patients on the device and admits the next ones of the list into the slots that
    finished patients free (screening + optimiser start while the others keep stepping).  Eleven patients through 3 resident slots
    (and through 4 in two alternating groups, and 5 walked in list order, and with a shared work counter): every patient's files are
    byte-identical to its single-patient run -- patients are independent (ref: one process per patient, main_one_train.cpp:41-324).
    Sizes reach three 64-blocks, where the default factorisation route depends on the batch: --pin-route on both sides."""
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-s", "-C", HOST, "medgp_train"])
    pans = [f"P{k:03d}" for k in range(11)]
    Ns = [60, 150, 48, 3, 64, 33, 170, 75, 129, 52, 90]          # P003: too few samples, never takes a slot
    opt = {"top_iteration_num": 6 if prior_index == 2 else 25}
    ref = make_experiment(tmp_path / "ref", pans, D=2, Q=3, R=2, N=Ns, prior_index=prior_index, opt=opt)
    for pan in pans:
        run(["--cfg", ref["cfg"], "--pan", pan, "--thread", "1", "--pin-route"])
    want = {f: open(os.path.join(ref["dirs"]["train"], f), "rb").read() for f in sorted(os.listdir(ref["dirs"]["train"]))}
    assert len(want) == 10 * (5 if prior_index == 2 else 4) + 2
    variants = (("a", ["--resident", "3"]), ("b", ["--resident", "4", "--pingpong-min", "2", "--host-threads", "3"]),
                ("c", ["--resident", "5", "--order", "list", "--admit-min", "2"]), ("d", ["--resident", "2", "--queue", str(tmp_path / "q.cnt")]))
    for tag, extra in variants:
        ex = make_experiment(tmp_path / tag, pans, D=2, Q=3, R=2, N=Ns, prior_index=prior_index, opt=opt)
        plist = tmp_path / f"pans_{tag}.txt"
        plist.write_text("\n".join(pans) + "\n")
        out = run(["--cfg", ex["cfg"], "--pan-list", str(plist), "--pin-route"] + extra)
        m = re.search(r"continuous admission: (\d+) patients through (\d+) resident slots in (\d+) admissions", out)
        assert m and int(m.group(1)) == 11 and int(m.group(2)) == int(extra[1]) and int(m.group(3)) >= 3, out[-1500:]
        got = {f: open(os.path.join(ex["dirs"]["train"], f), "rb").read() for f in sorted(os.listdir(ex["dirs"]["train"]))}
        assert sorted(got) == sorted(want), tag
        for f in want:
            assert got[f] == want[f], (tag, f)
        # the closing lines come in list order whatever the walk order was
        fin = re.findall(r"^finish individual id: (\S+) w/", out, flags=re.M)
        assert fin == pans, (tag, fin)
    assert int(open(tmp_path / "q.cnt").read()) >= 11          # the shared counter was walked to the end
    # a work counter that cannot be opened must not end as a success with nothing trained
    ex = make_experiment(tmp_path / "bad", pans[:3], D=2, Q=3, R=2, N=Ns[:3], prior_index=prior_index, opt=opt)
    plist = tmp_path / "pans_bad.txt"
    plist.write_text("\n".join(pans[:3]) + "\n")
    r = subprocess.run([EXE, "--cfg", ex["cfg"], "--pan-list", str(plist), "--queue", str(tmp_path / "no" / "such" / "dir" / "q.cnt")],
                       capture_output=True, text=True, timeout=180)
    assert r.returncode != 0 and "work counter" in r.stdout


@pytest.mark.parametrize("kernel_index", [0, 8])
def test_single_output_families_through_both_hosts(tmp_path, built_lib, kernel_index):
    """The reference's mains also run the single-output families (run_model_SE / run_model_SM, ref: main_one_train.cpp:120-152,
    main_one_test.cpp:144-186).  End to end through the C++ hosts on the device: screening arg-min and the SCG run of medgp_train
    against the oracle optimiser driven by the oracle objective (bounds in the order config.py:67-100 writes them, random initial
    hypers of get_hyp_SE / get_hyp_SM), cohort list == single runs byte for byte; then medgp_test (both passes, cohort list) with
    the trained hypers of the first patient as the mode kernel against the restated imputation loop."""
    from oracle import optimizer_oracle as OO
    from test_test_host_gpu import reference_loop_generic
    for exe in (EXE, os.path.join(HOST, "medgp_test"), os.path.join(HOST, "host_logic_test")):
        if not os.path.exists(exe):
            subprocess.check_call(["make", "-s", "-C", HOST, os.path.basename(exe)])
    logic = os.path.join(HOST, "host_logic_test")
    pans, Ns = ["S001", "S002"], [52, 70]
    Q = 1 if kernel_index == 0 else 3
    H = 3 if kernel_index == 0 else 1 + 3 * Q
    ex = make_experiment(tmp_path / "e", pans, Q=Q, N=Ns, kernel_index=kernel_index, opt={"top_iteration_num": 30, "online_learn_rate": 1e-4})
    ex2 = make_experiment(tmp_path / "c", pans, Q=Q, N=Ns, kernel_index=kernel_index, opt={"top_iteration_num": 30, "online_learn_rate": 1e-4})
    for pan in pans:
        run(["--cfg", ex["cfg"], "--pan", pan, "--thread", "1"])
    plist = tmp_path / "pans.txt"
    plist.write_text("\n".join(pans) + "\n")
    run(["--cfg", ex2["cfg"], "--pan-list", str(plist)])
    hb = tmp_path / "hyp.bin"
    subprocess.check_call([logic, "hyp", ex["cfg"], str(hb)], stdout=subprocess.DEVNULL)
    inits = np.fromfile(hb, np.float64).reshape(ex["opt"]["random_init_num"], H)
    data = {}
    for pan in pans:
        for name in ("train_init_hyp_", "train_hyp_"):
            a = np.fromfile(os.path.join(ex["dirs"]["train"], name + pan + ".bin"), np.float64)
            b = np.fromfile(os.path.join(ex2["dirs"]["train"], name + pan + ".bin"), np.float64)
            assert a.size == H and np.array_equal(a, b), (name, pan)
        db = tmp_path / f"data_{pan}.bin"
        subprocess.check_call([logic, "data", ex["cfg"], pan, str(db)], stdout=subprocess.DEVNULL)
        raw = open(db, "rb").read()
        n = int(np.frombuffer(raw, np.int32, 1)[0])
        t = np.frombuffer(raw, np.float32, n, 4 + 4 * n).copy()
        y = np.frombuffer(raw, np.float32, n, 4 + 8 * n).copy()
        data[pan] = (t, y)
        best, best_init = np.inf, None
        for th in inits:
            r = O.nlml_grad(kernel_index, Q, 1, 0, None, t, y, th, flag_grad=False)
            assert r["ok"]
            if r["nlml"] < best:
                best, best_init = r["nlml"], th
        assert np.array_equal(np.fromfile(os.path.join(ex["dirs"]["train"], "train_init_hyp_" + pan + ".bin"), np.float64), best_init), pan

        def obj(th):
            r = O.nlml_grad(kernel_index, Q, 1, 0, None, t, y, np.asarray(th, np.float64), flag_grad=True)
            return (True, r["nlml"], list(r["grad"])) if r["ok"] else (False, 0.0, [])
        loss, theta, _ = OO.scg(-30, best_init, obj)
        got = np.fromfile(os.path.join(ex["dirs"]["train"], "train_hyp_" + pan + ".bin"), np.float64)
        err = np.abs(got - np.array(theta)) / np.maximum(1.0, np.abs(theta))
        assert err.max() <= 1e-6, (pan, float(err.max()))
    # ---- medgp_test with the first patient's trained hypers as the mode kernel
    mode = np.fromfile(os.path.join(ex["dirs"]["train"], "train_hyp_S001.bin"), np.float64)
    fold_dir = os.path.join(ex["dirs"]["kernel"], "fold0")
    os.makedirs(fold_dir)
    open(os.path.join(fold_dir, "gmm_mode_mixture_num.txt"), "w").write(f"{Q}\n")
    mode.tofile(os.path.join(fold_dir, "gmm_mode_param.bin"))
    r = subprocess.run([os.path.join(HOST, "medgp_test"), "--cfg", ex["cfg"], "--pan-list", str(plist), "--fold", "0", "--kernclust-alg", "gmm"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    for pan in pans:
        t, y = data[pan]
        for flag_update, mode_name in ((False, "mean_wo_update"), (True, "mean_w_update")):
            feat, ci, et, err, pred = reference_loop_generic(kernel_index, Q, 1, 0, None, t, y, mode, flag_update, 1e-4, 0.9, ex["feature_index"])
            pre = os.path.join(ex["dirs"]["test"], f"test_{mode_name}_")
            gp = np.fromfile(pre + f"pred_{pan}.bin", np.float64)
            ge = np.fromfile(pre + f"error_{pan}.bin", np.float64)
            assert gp.size == t.size
            np.testing.assert_allclose(gp, pred, rtol=2e-5, atol=2e-6)
            np.testing.assert_allclose(ge, err, rtol=2e-5, atol=2e-6)


def test_train_skips_an_unreadable_patient_and_keeps_going(tmp_path, built_lib):
    """Advisor finding (round 5): one unreadable patient file -- or a patient larger than the list announced -- used to end the whole
    long-lived trainer in the middle of the run, dropping every resident patient's progress.  Now such a patient is skipped like one with
    too few samples (flag 0 through finish(), ref: main_one_train.cpp:185-201 for the flag files), everybody else is trained to the same
    bytes as without it, and the exit status is non-zero at the end (what medgp_test does with n_unreadable)."""
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-s", "-C", HOST, "medgp_train"])
    pans = [f"P{k:03d}" for k in range(6)]
    Ns = [60, 75, 48, 52, 64, 70]
    ref = make_experiment(tmp_path / "ref", pans, D=2, Q=3, R=2, N=Ns, prior_index=2)
    bad = make_experiment(tmp_path / "bad", pans, D=2, Q=3, R=2, N=Ns, prior_index=2)
    os.remove(os.path.join(bad["dirs"]["data"], "P002", "feature19.txt"))        # unreadable: a feature file is missing
    lists = {}
    for tag, ex in (("ref", ref), ("bad", bad)):
        plist = tmp_path / f"pans_{tag}.txt"
        # the list announces the sizes; P001's is too small in the `bad` run (the files hold 75 observations, the capacity is 70)
        plist.write_text("".join(f"{p} {n if not (tag == 'bad' and p == 'P001') else 20}\n" for p, n in zip(pans, Ns)))
        lists[tag] = str(plist)
    out_ref = run(["--cfg", ref["cfg"], "--pan-list", lists["ref"], "--resident", "3", "--pin-route"])
    r = subprocess.run([EXE, "--cfg", bad["cfg"], "--pan-list", lists["bad"], "--resident", "3", "--pin-route", "--max-n", "70"],
                       capture_output=True, text=True, timeout=240)
    assert r.returncode != 0, r.stdout[-2000:]
    assert "patient P002 skipped" in r.stdout and "more than the" in r.stdout and "2 patient(s) could not be read" in r.stdout
    assert "lock-step batches" in out_ref and "lock-step batches" in r.stdout
    for pan in pans:
        fa, fb = ref["dirs"]["train"], bad["dirs"]["train"]
        if pan in ("P002", "P001"):
            assert open(os.path.join(fb, f"train_flag_{pan}.txt")).read() == "0\n"
            assert not os.path.exists(os.path.join(fb, f"train_hyp_{pan}.bin"))
            continue
        for name in ("train_init_hyp_", "train_hyp_", "train_var_hyp_"):
            a = open(os.path.join(fa, name + pan + ".bin"), "rb").read()
            assert a and a == open(os.path.join(fb, name + pan + ".bin"), "rb").read(), (name, pan)
        assert open(os.path.join(fb, f"train_flag_{pan}.txt")).read() == "1\n"
