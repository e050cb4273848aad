"""bench.py's multi-rank path on ONE GPU box: two ranks launched exactly as the driver launches them
(python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 ...); with fewer GPUs than ranks the
ranks share device 0 and the process group is gloo (bench.py's fallback), so everything except RCCL itself is exercised:
sharding (weak and strong / LPT), max-over-ranks timing, whole-job value, ranks_seen, and -- the property that makes patient
sharding legitimate -- per-patient results identical to the single-rank run."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--steps", "2", "--warmup", "1", "--obs", "128", "--no-extra", "--no-cpu-baseline"]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(world, extra, dump, self_launch=False):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if world == 1 or self_launch:   # self_launch (round 6): `python bench.py --gpus N` with no launcher starts its N ranks itself
        env = {k: v for k, v in env.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world)]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world)]
    r = subprocess.run(cmd + COMMON + extra + ["--dump-results", dump], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout          # ONE JSON line, from rank 0
    res = {}
    for k in range(world):
        d = np.load(f"{dump}.rank{k}.npz")
        for g, nl, gs, st in zip(d["gids"], d["nlml"], d["gsum"], d["status"]):
            assert int(g) not in res
            res[int(g)] = (nl, gs, int(st))
    return json.loads(lines[0]), res


@pytest.mark.parametrize("mode", ["weak", "strong", "weak-self-launch"])
def test_two_ranks_on_one_gpu_match_one_rank(tmp_path, mode):
    if mode.startswith("weak"):
        one, r1 = _run(1, ["--patients", "48"], str(tmp_path / "one"))
        two, r2 = _run(2, ["--patients", "24"], str(tmp_path / "two"), self_launch=mode.endswith("self-launch"))
        mode = "weak"
    else:
        one, r1 = _run(1, ["--scaling", "strong", "--cohort", "50"], str(tmp_path / "one"))
        two, r2 = _run(2, ["--scaling", "strong", "--cohort", "50"], str(tmp_path / "two"))
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["scaling"] == mode
    assert np.isfinite(two["value"]) and two["value"] > 0 and two["unit"] == "evals/s"
    assert [r["rank"] for r in two["ranks_seen"]] == [0, 1] and two["backend"] in ("gloo", "nccl")
    assert sum(r["patients"] for r in two["ranks_seen"]) == two["config"]["cohort"] == len(r2) == len(r1)
    # value = whole-job units / max-over-ranks time
    assert abs(two["value"] - two["config"]["cohort"] * two["steps"] / (two["ms_per_step"] * 1e-3 * two["steps"])) <= 1e-6 * two["value"]
    assert sorted(r1) == sorted(r2)
    for g in r1:
        assert r1[g][2] == r2[g][2] == 0
        assert r1[g][0] == r2[g][0] and r1[g][1] == r2[g][1], (g, r1[g], r2[g])   # same bits whichever rank / batch evaluated the patient
    # round 5: the line shows every rank's own time (an imbalance is visible), and the whole-job time is the slowest rank's
    ms = [r["ms_per_step"] for r in two["ranks_seen"]]
    assert two["rank_ms_per_step"]["min"] == min(ms) and two["rank_ms_per_step"]["max"] == max(ms)
    assert max(ms) <= two["ms_per_step"] * 1.0001 and min(ms) > 0


def test_backend_switch_and_nccl_refuses_shared_gpus(tmp_path):
    """--backend gloo forces the CPU process group; --backend nccl with two ranks on this one-GPU box must FAIL LOUDLY (exit code,
    message) instead of quietly timing two ranks on one device."""
    import torch
    two, _ = _run(2, ["--patients", "8", "--backend", "gloo"], str(tmp_path / "g"))
    assert two["backend"] == "gloo"
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer GPUs than ranks")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--patients", "8", "--backend", "nccl"] + COMMON
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode != 0
    assert "needs one GPU per rank" in r.stderr + r.stdout
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_one_rank_under_torchrun_runs_the_rccl_path(tmp_path):
    """The nccl (= RCCL) branch of bench.py on the one GPU of the test box: launched by torch.distributed.run with ONE rank the process
    group is still created, so RCCL initialisation, the barriers, the max-time all-reduce on a device tensor and the all-gather of
    `ranks_seen` all execute -- the code path of the driver's 2 / 4 / 8-GPU runs, which no test could reach before round 5.  Results
    equal the plain single-process run bit for bit."""
    plain, r0 = _run(1, ["--patients", "16"], str(tmp_path / "plain"))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    dump = str(tmp_path / "tr")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--patients", "16", "--dump-results", dump] + COMMON
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["backend"] == "nccl" and line["n_gpus"] == 1 and len(line["ranks_seen"]) == 1
    assert line["nccl_version"] and line["nccl_version"][0].isdigit()          # RCCL reported its version: it was really initialised
    assert plain["backend"] == "none" and plain["nccl_version"] is None
    d = np.load(f"{dump}.rank0.npz")
    for g, nl, gs, st in zip(d["gids"], d["nlml"], d["gsum"], d["status"]):
        assert r0[int(g)] == (nl, gs, int(st))


def test_gpus_flag_must_match_the_launcher(tmp_path):
    """`--gpus 1` (or none: the default) under a two-rank launcher: exit status 2 and NO line -- a run can no longer report
    n_gpus != --gpus (round 5 parsed the flag and never read it)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--patients", "8"] + COMMON
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "refusing to print a line" in r.stderr
