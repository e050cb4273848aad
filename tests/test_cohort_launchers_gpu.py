"""GPU tests of the cohort launchers with the REAL hosts: medgp_amd.train_cohort (static shards vs the shared patient queue) and
medgp_amd.impute_cohort under torch.distributed.run with two gloo ranks that share the one GPU of the test box.  Patients are
independent (ref: medgpc/util/run_exp_generator.py:213-260 fans them out as scheduler jobs), so whichever rank and chunk
handles a patient, its files must be the same bytes as a single-process run."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from exp_fixture import make_experiment
from medgp_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "medgp_amd", "host")


def _launch(module, args, port):
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), "-m", module] + args, env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r


def _files(d, prefix):
    return {f: open(os.path.join(d, f), "rb").read() for f in sorted(os.listdir(d)) if f.startswith(prefix) and not f.endswith(".log")}


def test_train_and_test_cohort_launchers_two_ranks_one_gpu(tmp_path, built_lib):
    for exe in ("medgp_train", "medgp_test"):
        if not os.path.exists(os.path.join(HOST, exe)):
            subprocess.check_call(["make", "-s", "-C", HOST, exe])
    pans = [f"P{k:03d}" for k in range(6)]
    Ns = [40, 66, 52, 30, 70, 45]
    Q, D, R = 2, 2, 2
    exs = {}
    for tag in ("single", "static", "dynamic"):
        exs[tag] = make_experiment(tmp_path / tag, pans, D=D, Q=Q, R=R, N=Ns, prior_index=0, opt={"top_iteration_num": 12, "online_learn_rate": 1e-4})
    plist = tmp_path / "pans.txt"
    plist.write_text("\n".join(pans) + "\n")
    # one process, all patients: the reference result
    r = subprocess.run([os.path.join(HOST, "medgp_train"), "--cfg", exs["single"]["cfg"], "--pan-list", str(plist)], capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:]
    want = _files(exs["single"]["dirs"]["train"], "train_")
    assert len(want) == 6 * 4        # init hyp, hyp, num, flag per patient (prior mode 0: no var hyp)
    for tag, extra, port in (("static", ["--schedule", "static"], 29561), ("dynamic", ["--schedule", "dynamic", "--resident", "2"], 29563)):
        _launch("medgp_amd.train_cohort", ["--cfg", exs[tag]["cfg"], "--pan-list", str(plist), "--backend", "gloo", "--gather"] + extra, port)
        got = _files(exs[tag]["dirs"]["train"], "train_")
        got = {k: v for k, v in got.items() if not k.endswith(".busy")}
        assert sorted(got) == sorted(want), tag
        for f in want:
            assert got[f] == want[f], (tag, f)
        allrows = np.load(os.path.join(exs[tag]["dirs"]["train"], "cohort_train_hyp.npy"))
        assert allrows.shape[0] == 6 and np.all(allrows[:, 1] == 1)
        busy = [open(os.path.join(exs[tag]["dirs"]["train"], f"train_rank{k}.busy")).read().split() for k in range(2)]
        # (the shared queue hands patients to whichever trainer asks first: with six tiny patients one rank may get most of them)
        assert int(busy[0][1]) + int(busy[1][1]) == 6 and (tag == "dynamic" or (int(busy[0][1]) > 0 and int(busy[1][1]) > 0))
    # ---- the test launcher: mode kernel = one patient's trained hypers; two ranks vs one process
    mode = np.fromfile(os.path.join(exs["single"]["dirs"]["train"], "train_hyp_P001.bin"), np.float64)
    for tag in ("single", "static"):
        fold_dir = os.path.join(exs[tag]["dirs"]["kernel"], "fold0")
        os.makedirs(fold_dir)
        open(os.path.join(fold_dir, "gmm_mode_mixture_num.txt"), "w").write(f"{Q}\n")
        mode.tofile(os.path.join(fold_dir, "gmm_mode_param.bin"))
    r = subprocess.run([os.path.join(HOST, "medgp_test"), "--cfg", exs["single"]["cfg"], "--pan-list", str(plist), "--fold", "0", "--kernclust-alg", "gmm"],
                       capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:]
    want = _files(exs["single"]["dirs"]["test"], "test_mean_")
    assert len(want) == 6 * 2 * 6
    _launch("medgp_amd.impute_cohort", ["--cfg", exs["static"]["cfg"], "--pan-list", str(plist), "--fold", "0", "--kernclust-alg", "gmm", "--backend", "gloo"], 29565)
    got = _files(exs["static"]["dirs"]["test"], "test_mean_")
    assert sorted(got) == sorted(want)
    for f in want:
        assert got[f] == want[f], f


def test_rccl_branches_with_one_rank(tmp_path, built_lib):
    """The nccl (= RCCL) branches of the three launchers on the one GPU of the test box: ONE rank under torch.distributed.run with
    MEDGP_FORCE_COLLECTIVES=1 creates the process group and runs the all-reduce of the exit status, the all-gather of the trained
    hypers (train_cohort --gather), the padded all-gather of the KDE modes (cohort_mode) and the barriers on device tensors -- the
    code the multi-GPU launch runs, which only gloo stand-ins reached before round 5."""
    import torch
    if torch.cuda.device_count() < 1:
        pytest.skip("needs a GPU")
    for exe in ("medgp_train", "medgp_test"):
        if not os.path.exists(os.path.join(HOST, exe)):
            subprocess.check_call(["make", "-s", "-C", HOST, exe])
    pans = [f"P{k:03d}" for k in range(4)]
    Q, D, R = 2, 2, 2
    ex = make_experiment(tmp_path / "e", pans, D=D, Q=Q, R=R, N=[40, 66, 52, 30], prior_index=0, opt={"top_iteration_num": 8, "online_learn_rate": 1e-4})
    plist = tmp_path / "pans.txt"
    plist.write_text("\n".join(pans) + "\n")
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", MEDGP_FORCE_COLLECTIVES="1")

    def launch(args, port):
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                            "--master-port", str(port)] + args, env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        return r
    launch(["-m", "medgp_amd.train_cohort", "--cfg", ex["cfg"], "--pan-list", str(plist), "--backend", "nccl", "--gather"], 29571)
    rows = np.load(os.path.join(ex["dirs"]["train"], "cohort_train_hyp.npy"))
    assert rows.shape[0] == 4 and np.all(rows[:, 1] == 1)
    for k, pan in enumerate(pans):
        assert np.array_equal(rows[k, 2:], np.fromfile(os.path.join(ex["dirs"]["train"], f"train_hyp_{pan}.bin"), np.float64))
    fold_dir = os.path.join(ex["dirs"]["kernel"], "fold0")
    os.makedirs(fold_dir)
    open(os.path.join(fold_dir, "gmm_mode_mixture_num.txt"), "w").write(f"{Q}\n")
    np.fromfile(os.path.join(ex["dirs"]["train"], "train_hyp_P001.bin"), np.float64).tofile(os.path.join(fold_dir, "gmm_mode_param.bin"))
    launch(["-m", "medgp_amd.impute_cohort", "--cfg", ex["cfg"], "--pan-list", str(plist), "--fold", "0", "--kernclust-alg", "gmm", "--backend", "nccl"], 29573)
    assert open(os.path.join(ex["dirs"]["test"], "test_mean_w_update_flag_P002.txt")).read() == "1\n"
    r = launch([os.path.join(ROOT, "tests", "rccl_cohort_mode_worker.py")], 29575)
    assert "RCCL_COHORT_MODE_OK" in r.stdout
