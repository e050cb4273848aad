#!/usr/bin/env python3
"""Cohort online-imputation test across the GPUs of one node: one process per GPU (torch.distributed), the test patients
of a cross-validation fold are partitioned with the LPT rule (no data-path collective -- patients are independent, the
reference fans them out as one scheduler job per patient, ref: scripts/test_della.sh:46,
medgpc/util/run_exp_generator.py:213-260), every rank runs the lock-step tester
`medgp_test --pan-list <its shard> --fold k --kernclust-alg <alg> --device <local rank>`.

Cost model of a test patient (what the LPT rule balances): the update pass factorises, for every imputed observation, the
observations of the preceding 72 h plus the other observations of its time stamp (ref: main_one_test.cpp:287-300, :358-365),
so cost = sum over observations of N_tt^3, plus n^3 for the one shared factorisation of the no-update pass.

The only collective is the agreement on the exit status (all_reduce MAX): every rank returns non-zero if any shard failed.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \
        -m medgp_amd.impute_cohort --cfg exp_setup.json --pan-list pans.txt --fold 0 --kernclust-alg gmm
"""
import argparse
import json
import os
import subprocess
import sys

import numpy as np

from . import shard

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_EXE = os.path.join(HERE, "host", "medgp_test")


def read_times(cfg, pan):
    """All observation times of a patient from its feature files (count, then t / v pairs, ref dataio/c_experiment.cpp:296-307)."""
    ts = []
    for fi in cfg["feature_index"].split():
        try:
            with open(os.path.join(cfg["data_dir"], pan, f"feature{fi}.txt")) as f:
                tok = f.read().split()
            n = int(float(tok[0]))
            ts.append(np.array(tok[1:1 + 2 * n:2], dtype=np.float32))
        except (OSError, ValueError, IndexError):
            pass
    return np.concatenate(ts) if ts else np.zeros(0, np.float32)


def impute_cost(t, window=72.0):
    """sum_i N_i^3 + n^3: N_i = observations strictly before t_i within the window + the other observations at t_i."""
    t = np.sort(np.asarray(t, dtype=np.float64))
    n = t.size
    if n == 0:
        return 0.0
    lo = np.searchsorted(t, t - window, side="left")
    first_same = np.searchsorted(t, t, side="left")
    last_same = np.searchsorted(t, t, side="right")
    ni = (first_same - lo) + (last_same - first_same - 1)
    return float(np.sum(ni.astype(np.float64) ** 3) + float(n) ** 3)


def lpt(costs, world):
    """Longest processing time first; deterministic (ties by index). Returns a list of sorted index lists."""
    costs = np.asarray(costs, dtype=np.float64)
    order = np.lexsort((np.arange(costs.size), -costs))
    loads = np.zeros(world)
    parts = [[] for _ in range(world)]
    for i in order:
        r = int(np.argmin(loads))
        parts[r].append(int(i))
        loads[r] += costs[i]
    return [sorted(p) for p in parts]


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", required=True)
    ap.add_argument("--pan-list", required=True)
    ap.add_argument("--fold", type=int, default=0)
    ap.add_argument("--kernclust-alg", required=True)
    ap.add_argument("--exe", default=DEFAULT_EXE)
    ap.add_argument("--backend", default=None, help="nccl (default with GPUs) or gloo")
    ap.add_argument("--max-batch", type=int, default=0)
    ap.add_argument("--timeout-hours", type=float, default=48.0)
    args = ap.parse_args(argv)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    try:                       # more ranks than GPUs (tests on a one-GPU box): ranks share devices round robin
        import torch
        ndev = torch.cuda.device_count()       # (counting devices does not initialise the GPU)
    except Exception:          # noqa: BLE001
        ndev = 0
    device = local_rank % ndev if ndev > 0 else local_rank

    cfg = json.load(open(args.cfg))
    pans = [p for p in open(args.pan_list).read().split() if p]
    costs = [impute_cost(read_times(cfg, p)) for p in pans]
    parts = lpt(costs, world)
    mine = [pans[i] for i in parts[rank]]
    rc = 0
    if mine:
        shard_file = os.path.join(cfg["exp_test_dir"], f"pan_shard_fold{args.fold}_rank{rank}.txt")
        with open(shard_file, "w") as f:
            f.write("\n".join(mine) + "\n")
        cmd = [args.exe, "--cfg", args.cfg, "--pan-list", shard_file, "--fold", str(args.fold), "--kernclust-alg", args.kernclust_alg,
               "--device", str(device)]
        if args.max_batch > 0:
            cmd += ["--max-batch", str(args.max_batch)]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        rc = shard.exit_status(r.returncode)
        with open(os.path.join(cfg["exp_test_dir"], f"test_fold{args.fold}_rank{rank}.log"), "w") as f:
            f.write(r.stdout)
    # MEDGP_FORCE_COLLECTIVES=1: also with ONE rank (under torch.distributed.run): lets a one-GPU box run the RCCL code path of the
    # multi-GPU launch -- process group, all-reduce of the exit status, all-gather, barrier (tests/test_cohort_launchers_gpu.py)
    if world > 1 or (os.environ.get("MEDGP_FORCE_COLLECTIVES") == "1" and "MASTER_ADDR" in os.environ):
        # created only now, after the shard has been processed (ranks can arrive far apart; see train_cohort.py)
        import datetime
        import torch
        import torch.distributed as dist
        backend = args.backend or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(device)      # (the device the trainer ran on: local_rank % ndev, ranks may share GPUs)
        dist.init_process_group(backend, timeout=datetime.timedelta(hours=args.timeout_hours))
        dev = torch.device("cuda", device) if dist.get_backend() == "nccl" else torch.device("cpu")
        flag = torch.tensor([rc], dtype=torch.int64, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        rc = int(flag.item())
        dist.barrier()
        dist.destroy_process_group()
    return rc


if __name__ == "__main__":
    sys.exit(main())
