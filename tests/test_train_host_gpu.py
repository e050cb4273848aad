"""GPU test of medgp_train (the main_one_train replacement): file surface, lock-step cohort training ==
one-patient-at-a-time training bit for bit, losses decrease, prior-mode-2 state is written."""
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
EXE = os.path.join(HOST, "medgp_train")


def run(args, timeout=600):
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
