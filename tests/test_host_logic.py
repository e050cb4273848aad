"""Host-side logic: synthetic cohort generator, sharding, bench aggregation; world_size-2 gloo run."""
import os
import subprocess
import sys

import numpy as np

from medgp_amd import shard, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_synth_shapes_and_determinism():
    m, t, y = synth.patient(1, 3, 24, 512)
    m2, t2, y2 = synth.patient(1, 3, 24, 512)
    assert m.dtype == np.int32 and t.dtype == np.float32 and y.dtype == np.float32
    np.testing.assert_array_equal(t, t2)
    np.testing.assert_array_equal(y, y2)
    assert np.all(np.diff(m) >= 0) and np.bincount(m, minlength=24).tolist() == [22] * 8 + [21] * 16
    for d in range(24):
        assert np.all(np.diff(t[m == d]) >= 0)
    assert len(np.unique(t)) < t.size           # shared draw times exist
    th = synth.theta(1, 3, 7, 5, 24, 8)
    assert th.shape == (1114,) and synth.num_hyp(7, 5, 24, 8) == 1114
    assert synth.num_hyp(7, 5, 2, 2) == 42 and synth.num_hyp(7, 5, 64, 8) == 2954
    mi, ti, yi = synth.patient(1, 3, 24, 512, interleave=True)
    assert not np.all(np.diff(mi) >= 0) and sorted(ti.tolist()) == sorted(t.tolist())


def test_lpt_partition_balanced_and_complete():
    rng = np.random.default_rng(0)
    ns = rng.integers(20, 900, size=203)
    parts = shard.lpt_partition(ns, 8)
    allidx = np.sort(np.concatenate(parts))
    np.testing.assert_array_equal(allidx, np.arange(203))
    loads = np.array([shard.cost(ns[p]).sum() for p in parts])
    assert loads.max() / loads.mean() < 1.05
    again = shard.lpt_partition(ns, 8)
    for a, b in zip(parts, again):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(shard.weak_shard(512, 3), np.arange(1536, 2048))


def test_two_rank_gloo_sharded_evaluation_matches_single_process():
    """N>1 path on CPU: two gloo ranks evaluate disjoint shards (through the oracle, the only CPU evaluator
    there is -- test infrastructure), rank 0 gathers; results must be bit-identical to one process."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29541",
                          os.path.join(ROOT, "tests", "gloo_shard_worker.py")],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "GLOO_SHARD_OK" in out.stdout


def test_train_cohort_launcher_shards_and_gathers_over_gloo(tmp_path):
    """medgp_amd.train_cohort on 2 CPU ranks (gloo) with a stand-in trainer: LPT sharding of the patient list,
    one trainer invocation per rank, all-gather of the trained hypers into cohort_train_hyp.npy."""
    import json
    import zlib
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from exp_fixture import make_experiment
    pans = [f"P{k:03d}" for k in range(7)]
    ex = make_experiment(tmp_path, pans, D=2, Q=2, R=2, N=[20, 31, 24, 40, 22, 35, 28])
    plist = tmp_path / "pans.txt"
    plist.write_text("\n".join(pans) + "\n")
    env = dict(os.environ, PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29547", "-m", "medgp_amd.train_cohort",
                          "--cfg", ex["cfg"], "--pan-list", str(plist), "--gather", "--backend", "gloo", "--schedule", "static",
                          "--exe", os.path.join(ROOT, "tests", "gloo_train_cohort_worker.sh")],
                         env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stdout + out.stderr
    got = np.load(os.path.join(ex["dirs"]["train"], "cohort_train_hyp.npy"))
    H = 2 + 2 * (2 * 2 + 2 + 2)
    assert got.shape == (7, H + 2) and np.array_equal(got[:, 0], np.arange(7)) and np.all(got[:, 1] == 1)
    for k, pan in enumerate(pans):
        exp = np.random.default_rng(zlib.crc32(pan.encode())).normal(size=H)
        assert np.array_equal(got[k, 2:], exp)
    shards = [[ln.split()[0] for ln in open(os.path.join(ex["dirs"]["train"], f"pan_shard_rank{r}.txt"))] for r in range(2)]
    assert sorted(shards[0] + shards[1]) == pans and shards[0] and shards[1]


def test_train_cohort_dynamic_queue_balances_skewed_budgets_over_gloo(tmp_path):
    """Ranks pull patients from a shared counter (one long-lived trainer per rank): with half of the patients ten times as
    expensive as the cost model thinks (evaluation budgets are not known up front: early stops, failed line searches), both ranks stay busy to the end --
    max / mean busy time <= 1.15 -- and the gathered result equals the static schedule's."""
    import shutil
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from exp_fixture import make_experiment
    pans = [f"P{k:03d}" for k in range(32)]
    ex = make_experiment(tmp_path, pans, D=2, Q=2, R=2, N=[20 + (k % 5) for k in range(32)])
    plist = tmp_path / "pans.txt"
    plist.write_text("\n".join(pans) + "\n")
    # the expensive half: every patient whose N is largest -- exactly the ones a longest-first order hands out first
    heavy = [p for k, p in enumerate(pans) if k % 5 >= 3] + pans[:3]
    env = dict(os.environ, PYTHONPATH=ROOT, MEDGP_FAKE_UNIT="0.03", MEDGP_FAKE_HEAVY=" ".join(heavy))
    results = {}
    for schedule, port in (("dynamic", "29551"), ("static", "29553")):
        for f in os.listdir(ex["dirs"]["train"]):
            os.remove(os.path.join(ex["dirs"]["train"], f))
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                              "--master-addr", "127.0.0.1", "--master-port", port, "-m", "medgp_amd.train_cohort",
                              "--cfg", ex["cfg"], "--pan-list", str(plist), "--gather", "--backend", "gloo", "--schedule", schedule,
                              "--exe", os.path.join(ROOT, "tests", "gloo_train_cohort_worker.sh")],
                             env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
        assert out.returncode == 0, out.stdout + out.stderr
        busy = [[float(v) for v in open(os.path.join(ex["dirs"]["train"], f"train_rank{r}.busy")).read().split()] for r in range(2)]
        results[schedule] = (np.load(os.path.join(ex["dirs"]["train"], "cohort_train_hyp.npy")), busy)
        assert int(busy[0][1] + busy[1][1]) == 32
    got, busy = results["dynamic"]
    np.testing.assert_array_equal(got, results["static"][0])          # same patients, same bytes, whoever trained them
    assert got.shape[0] == 32 and np.all(got[:, 1] == 1)
    t = np.array([b[0] for b in busy])
    assert t.max() / t.mean() <= 1.15, (busy, results["static"][1])
    assert not any(f.startswith(".patient_queue_") for f in os.listdir(ex["dirs"]["train"]))   # the counter is removed at the end
    # a trainer that dies on a signal (HIP fault -> SIGABRT, OOM kill -> SIGKILL: subprocess reports -9) must fail the WHOLE launch:
    # the negative code used to lose against the healthy rank's 0 in all_reduce(MAX) (advisor finding, round 4)
    for schedule, port in (("dynamic", "29555"), ("static", "29559")):
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                              "--master-addr", "127.0.0.1", "--master-port", port, "-m", "medgp_amd.train_cohort",
                              "--cfg", ex["cfg"], "--pan-list", str(plist), "--gather", "--backend", "gloo", "--schedule", schedule,
                              "--exe", os.path.join(ROOT, "tests", "gloo_train_cohort_worker.sh")],
                             env=dict(env, MEDGP_FAKE_UNIT="0", MEDGP_FAKE_KILL="P017"), capture_output=True, text=True, timeout=900, cwd=ROOT)
        assert out.returncode != 0, (schedule, out.stdout[-500:])
        assert not any(f.startswith(".patient_queue_") for f in os.listdir(ex["dirs"]["train"]))


def test_test_cohort_launcher_cost_model_and_gloo_run(tmp_path):
    """medgp_amd.impute_cohort: cost model = sum N_tt^3 over the 72-h windows + n^3; LPT shards over 2 gloo ranks with a stand-in
    tester; a failing shard makes every rank return non-zero."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from exp_fixture import make_experiment
    from medgp_amd import impute_cohort
    # three observations at t = 0, 1, 1 and one at t = 100: N_i = 0, 2, 2 (before + same time), 0 (window) -> 16 + 4^3
    assert impute_cohort.impute_cost(np.array([0.0, 1.0, 1.0, 100.0])) == 16.0 + 64.0
    assert impute_cohort.impute_cost(np.zeros(0)) == 0.0
    assert impute_cohort.lpt([5.0, 1.0, 4.0, 1.0], 2) == [[0, 3], [1, 2]]
    pans = [f"T{k:02d}" for k in range(5)]
    ex = make_experiment(tmp_path, pans, D=2, Q=2, R=2, N=[20, 44, 24, 40, 30])
    import json
    cfg = json.load(open(ex["cfg"]))
    assert impute_cohort.read_times(cfg, "T01").size == 44
    for names, want_rc in ((pans, 0), (pans[:4] + ["FAIL"], 3)):
        plist = tmp_path / "tpans.txt"
        plist.write_text("\n".join(names) + "\n")
        if "FAIL" in names:
            os.makedirs(os.path.join(ex["dirs"]["data"], "FAIL"), exist_ok=True)
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                              "--master-addr", "127.0.0.1", "--master-port", "29557", "-m", "medgp_amd.impute_cohort",
                              "--cfg", ex["cfg"], "--pan-list", str(plist), "--fold", "0", "--kernclust-alg", "gmm", "--backend", "gloo",
                              "--exe", os.path.join(ROOT, "tests", "gloo_test_cohort_worker.sh")],
                             env=dict(os.environ, PYTHONPATH=ROOT), capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert (out.returncode != 0) == (want_rc != 0), out.stdout + out.stderr
        shards = [open(os.path.join(ex["dirs"]["test"], f"pan_shard_fold0_rank{r}.txt")).read().split() for r in range(2)]
        assert sorted(shards[0] + shards[1]) == sorted(names) and shards[0] and shards[1]
        if want_rc == 0:
            for p in pans:
                assert open(os.path.join(ex["dirs"]["test"], f"test_mean_w_update_flag_{p}.txt")).read() == "1\n"


def test_bench_refuses_ranks_that_share_a_gpu_under_nccl():
    """bench.check_ranks: the first real 8-GPU run must fail loudly, not quietly, if two ranks land on one device."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    mk = lambda r, dev, uuid: {"rank": r, "device": dev, "name": "x", "uuid": uuid, "host": "h", "patients": 4, "ms_per_step": 1.0}
    ok = [mk(r, r, f"u{r}") for r in range(8)]
    assert bench.check_ranks(ok, 8, "nccl") is None
    dup = ok[:7] + [mk(7, 6, "u6")]
    assert "share a GPU" in bench.check_ranks(dup, 8, "nccl")
    assert bench.check_ranks(dup, 8, "gloo") is None                       # the CPU stand-in shares devices by design
    assert "of 8 ranks reported" in bench.check_ranks(ok[:7] + [None], 8, "nccl")
    assert "rank ids" in bench.check_ranks(ok[:7] + [mk(3, 7, "u7")], 8, "nccl")
    nouuid = [mk(r, r, "") for r in range(2)]                             # a torch without the uuid property: device index stands in
    assert bench.check_ranks(nouuid, 2, "nccl") is None
    assert "share a GPU" in bench.check_ranks([mk(0, 0, ""), mk(1, 0, "")], 2, "nccl")


def test_exit_status_of_a_signalled_child_is_positive():
    assert shard.exit_status(0) == 0 and shard.exit_status(3) == 3
    assert shard.exit_status(-9) == 137 and shard.exit_status(-6) == 134      # SIGKILL, SIGABRT: the shell's 128 + s
    assert max(0, shard.exit_status(-11)) > 0


def _load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


def test_bench_gpus_flag_means_something(capsys):
    """bench.resolve_launch: --gpus N must equal the launcher's WORLD_SIZE (exit 2, no line), and without a launcher N > 1 starts the
    N ranks itself through torch.distributed.run on 127.0.0.1 and leaves with that launch's exit code."""
    bench = _load_bench()
    env = {"RANK": "0", "WORLD_SIZE": "2", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29500"}
    assert bench.resolve_launch(2, env, []) is None                       # launcher and flag agree: go on in this process
    assert bench.resolve_launch(8, env, []) == 2                          # mismatch: refuse
    assert "refusing to print a line" in capsys.readouterr().err
    assert bench.resolve_launch(1, dict(env, WORLD_SIZE="2", RANK="1"), []) == 2
    assert capsys.readouterr().err == ""                                  # (only rank 0 speaks)
    assert bench.resolve_launch(1, {}, []) is None                        # plain `python bench.py`
    seen = []
    rc = bench.resolve_launch(4, {}, ["--gpus", "4", "--steps", "3"], run=lambda cmd: (seen.append(cmd), 7)[1])
    assert rc == 7
    cmd = seen[0]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and os.path.basename(cmd[-5]) == "bench.py"


def test_bench_gpus_2_without_a_launcher_never_prints_an_n_gpus_1_line():
    """`python bench.py --gpus 2` with no launcher environment (here: no GPU either, so the ranks fail): whatever happens, no JSON line
    with n_gpus != 2 comes out and the exit status is not 0."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-extra", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=300, env=env)
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert all(ln["n_gpus"] == 2 for ln in lines)
    import torch
    if not torch.cuda.is_available():
        assert r.returncode != 0 and not lines
    assert "starting 2 ranks" in r.stderr
