"""CPU tests of the C++ host logic (no GPU): config / data file formats and the random initial hypers
(glibc srand/rand, ref dataio/c_experiment.cpp:418-441,493-564), and the resumable SCG state machine."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from exp_fixture import make_experiment

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "medgp_amd", "host")
EXE = os.path.join(HOST, "host_logic_test")
REF_PI = 3.14159265


@pytest.fixture(scope="module")
def exe(built_lib):
    subprocess.check_call(["make", "-s", "-C", HOST, "host_logic_test"])
    return EXE


def test_scg_state_machine_matches_direct_form(exe):
    r = subprocess.run([exe, "scg"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "SCG_PASS" in r.stdout, r.stdout + r.stderr


def test_patient_loader_zscores_and_groups_by_output(exe, tmp_path):
    ex = make_experiment(tmp_path, ["P001"], D=2, N=41)
    out = tmp_path / "d.bin"
    r = subprocess.run([exe, "data", ex["cfg"], "P001", str(out)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
    b = open(out, "rb").read()
    n = np.frombuffer(b, np.int32, 1)[0]
    meta = np.frombuffer(b, np.int32, n, 4)
    t = np.frombuffer(b, np.float32, n, 4 + 4 * n)
    y = np.frombuffer(b, np.float32, n, 4 + 8 * n)
    em, et, ey = [], [], []
    for j in range(2):
        tt, vv = ex["raw"]["P001"][j]
        # the loader parses the 6-decimal text; reproduce that path exactly
        tt = np.array([np.float32(f"{a:.6f}") for a in tt], np.float32)
        vv = np.array([np.float32(f"{a:.6f}") for a in vv], np.float32)
        em += [j] * len(tt)
        et.append(tt)
        ey.append(((vv.astype(np.float64) - ex["stats"][j][0]) / ex["stats"][j][1]).astype(np.float32))
    assert n == len(em) and np.array_equal(meta, em)
    assert np.array_equal(t, np.concatenate(et)) and np.array_equal(y, np.concatenate(ey))


def test_random_initial_hypers_follow_glibc_rand(exe, tmp_path):
    ex = make_experiment(tmp_path, ["P001"], D=2, Q=3, R=2)
    out = tmp_path / "h.bin"
    r = subprocess.run([exe, "hyp", ex["cfg"], str(out)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
    Q, D, R, o = ex["Q"], ex["D"], ex["R"], ex["opt"]
    H = D + Q * (D * R + 2 + D)
    got = np.fromfile(out, np.float64).reshape(o["random_init_num"], H)
    libc = ctypes.CDLL("libc.so.6")
    libc.srand(o["random_seed"])

    def one(lb, ub, scale, inv, log):
        lb, ub = float("{:6.6f}".format(lb)), float("{:6.6f}".format(ub))
        temp = float(libc.rand() % 4096) + 1.0
        temp *= (ub - lb)
        temp = temp / 4096.0
        a = scale * (temp + lb)
        if inv:
            a = 1.0 / a
        return np.log(a) if log else a
    exp = np.empty_like(got)
    for i in range(o["random_init_num"]):
        h = []
        for _ in range(D):
            h.append(one(o["lower_bound_noise"], o["upper_bound_noise"], 1.0, False, True))
        for _ in range(Q * D * R):
            h.append(one(o["lower_bound_a"], o["upper_bound_a"], 0.9 / np.sqrt(float(Q) * float(R)), False, False))
        for _ in range(Q):
            h.append(np.log(1.0 / one(o["lower_bound_period"], o["upper_bound_period"], 1.0, False, False)))
        for _ in range(Q):
            h.append(np.log(1.0 / (2 * REF_PI * one(o["lower_bound_lengthscale"], o["upper_bound_lengthscale"], 1.0, False, False))))
        for _ in range(Q * D):
            h.append(one(o["lower_bound_lambda"], o["upper_bound_lambda"], 0.1 / float(Q), False, True))
        exp[i] = h
    np.testing.assert_allclose(got, exp, rtol=1e-15, atol=0)


def test_config_type_checks_mirror_rapidjson_asserts(exe, tmp_path):
    import json
    ex = make_experiment(tmp_path, ["P001"])
    cfg = json.load(open(ex["cfg"]))
    cfg["eta"] = 1          # an integer literal fails the reference's assert(d["eta"].IsFloat()) (c_experiment.cpp:105)
    bad = tmp_path / "bad.json"
    json.dump(cfg, open(bad, "w"))
    r = subprocess.run([exe, "hyp", str(bad), str(tmp_path / "x.bin")], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "eta" in r.stdout
