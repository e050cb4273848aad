#!/usr/bin/env python3
"""Cohort training across the GPUs of one node: one process per GPU (torch.distributed), the patient list is
partitioned with the LPT rule of medgp_amd.shard (no data-path collective -- patients are independent, the
reference fans them out as one SLURM job per patient, ref: medgpc/util/run_exp_generator.py:213-260), every rank
runs the lock-step trainer `medgp_train --pan-list <its shard> --device <local rank>`.

The ONE collective (optional, --gather): after training, the per-patient hyper vectors train_hyp_<PAN>.bin are
all-gathered (RCCL over xGMI under the nccl backend, gloo on CPU) into <exp_train_dir>/cohort_train_hyp.npy --
the input of the cohort-level kernel clustering step (ref: medgpc/util/binaryIO.py:20-35 read_train_kernel).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \
        -m medgp_amd.train_cohort --cfg exp_setup.json --pan-list pans.txt --gather
"""
import argparse
import json
import os
import subprocess
import sys

import numpy as np

from . import shard

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_EXE = os.path.join(HERE, "host", "medgp_train")


def count_observations(cfg, pan):
    """N of a patient from the feature files' headers (first line = count, ref dataio/c_experiment.cpp:296-299)."""
    n = 0
    for fi in cfg["feature_index"].split():
        try:
            with open(os.path.join(cfg["data_dir"], pan, f"feature{fi}.txt")) as f:
                n += int(float(f.readline().split()[0]))
        except (OSError, ValueError, IndexError):
            pass
    return n


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", required=True)
    ap.add_argument("--pan-list", required=True)
    ap.add_argument("--exe", default=DEFAULT_EXE)
    ap.add_argument("--gather", action="store_true")
    ap.add_argument("--backend", default=None, help="nccl (default with GPUs) or gloo")
    ap.add_argument("--max-batch", type=int, default=1024)
    ap.add_argument("--timeout-hours", type=float, default=48.0,
                    help="process-group timeout: how long a finished rank waits for the slowest shard")
    args = ap.parse_args(argv)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None

    cfg = json.load(open(args.cfg))
    pans = [p for p in open(args.pan_list).read().split() if p]
    ns = [count_observations(cfg, p) for p in pans]
    parts = shard.lpt_partition(ns, world, Q=int(cfg["Q"]))
    mine = [pans[i] for i in parts[rank]]
    rc = 0
    if mine:
        shard_file = os.path.join(cfg["exp_train_dir"], f"pan_shard_rank{rank}.txt")
        with open(shard_file, "w") as f:
            f.write("\n".join(mine) + "\n")
        r = subprocess.run([args.exe, "--cfg", args.cfg, "--pan-list", shard_file, "--device", str(local_rank),
                            "--max-batch", str(args.max_batch)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        rc = r.returncode
        with open(os.path.join(cfg["exp_train_dir"], f"train_rank{rank}.log"), "w") as f:
            f.write(r.stdout)
    if world > 1:
        # The process group is created only now, AFTER the training subprocess has returned: lock-step SCG runs until the
        # slowest patient of a shard finishes, so ranks can arrive hours apart, and a group created up front would have its
        # first collective (below) aborted by the default watchdog timeout (nccl 10 min).  The generous timeout covers the
        # rendezvous skew itself.
        import datetime
        import torch
        import torch.distributed as dist
        backend = args.backend or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend, timeout=datetime.timedelta(hours=args.timeout_hours))
        dev = torch.device("cuda", local_rank) if dist.get_backend() == "nccl" else torch.device("cpu")
        flag = torch.tensor([rc], dtype=torch.int64, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        rc = int(flag.item())
        if args.gather and rc == 0:
            H = int(cfg["D"]) + int(cfg["Q"]) * (int(cfg["D"]) * int(cfg["R"]) + 2 + int(cfg["D"])) if int(cfg["kernel_index"]) == 7 \
                else (1 + 3 * int(cfg["Q"]) if int(cfg["kernel_index"]) == 8 else 3)
            # fixed-size contribution per rank: rows of (global patient index, flag, theta); absent rows are NaN
            rows = max(len(p) for p in parts)
            buf = torch.full((rows, H + 2), float("nan"), dtype=torch.float64, device=dev)
            for k, gi in enumerate(parts[rank]):
                pan = pans[gi]
                fn = os.path.join(cfg["exp_train_dir"], f"train_hyp_{pan}.bin")
                buf[k, 0] = float(gi)
                if os.path.exists(fn):
                    th = np.fromfile(fn, np.float64)
                    if th.size == H:
                        buf[k, 1] = 1.0
                        buf[k, 2:] = torch.from_numpy(th).to(dev)
                        continue
                buf[k, 1] = 0.0
            out = [torch.empty_like(buf) for _ in range(world)]
            dist.all_gather(out, buf)   # the only collective on real data: <= P * H doubles
            if rank == 0:
                allrows = torch.cat(out).cpu().numpy()
                allrows = allrows[~np.isnan(allrows[:, 0])]
                allrows = allrows[np.argsort(allrows[:, 0])]
                np.save(os.path.join(cfg["exp_train_dir"], "cohort_train_hyp.npy"), allrows)
        dist.barrier()
        dist.destroy_process_group()
    return rc


if __name__ == "__main__":
    sys.exit(main())
