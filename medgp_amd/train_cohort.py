#!/usr/bin/env python3
"""Cohort training across the GPUs of one node: one process per GPU (torch.distributed), no data-path collective --
patients are independent; the reference fans them out as one scheduler job per patient and the scheduler balances them
dynamically (ref: medgpc/util/run_exp_generator.py:213-260, scripts/slurm_della.json:6-62).  Every rank runs ONE long-lived
lock-step trainer `medgp_train --pan-list <list> --device <local rank>` (continuous admission: the trainer keeps --resident
patients on its GPU and pulls the next one into every slot a finished patient frees, medgp_amd/host/medgp_train.cpp).

Scheduling (--schedule):
  dynamic (default)  every rank's trainer is given the WHOLE list (with the observation counts, so that all of them walk it in the
                     same order: longest patient first) and `--queue <file>`: the trainers PULL PATIENTS from one shared counter (a
                     flock-protected file in exp_train_dir: no process group exists while the trainers run).  The cost model cannot
                     know how many evaluations a patient will take -- the variational-EM loop stops early on a relative loss change
                     < 0.5 % (ref: util/c_optimizer_varEM.cpp:89-95), SCG line searches fail (ref: util/c_optimizer_scg.cpp:125-131)
                     -- so a static partition leaves ranks idle behind the slowest shard; with a queue a rank whose patients finish
                     early simply takes more.  (Round 4 pulled chunks of 256 patients and started a trainer process per chunk: every
                     chunk ended in small batches and paid 0.55 s of start-up.)  All ranks must be on one node (one file system lock).
  static             one LPT shard per rank, fixed before the first evaluation (rounds 1-3).
A patient's results do not depend on which rank trained it nor on who its batch-mates were, up to the last bits of the
factorisation route (same kernels for the same batch-size class); pass --pin-route to the trainer through --exe-args for
bit-identity across batch compositions as well.

The ONE collective on data (optional, --gather): after training, the per-patient hyper vectors train_hyp_<PAN>.bin are
all-gathered (RCCL over xGMI under the nccl backend, gloo on CPU) into <exp_train_dir>/cohort_train_hyp.npy --
the input of the cohort-level kernel clustering step (ref: medgpc/util/binaryIO.py:20-35 read_train_kernel).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \
        -m medgp_amd.train_cohort --cfg exp_setup.json --pan-list pans.txt --gather
"""
import argparse
import fcntl
import json
import os
import re
import subprocess
import sys
import time

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


def take_ticket(path):
    """Atomic fetch-and-increment of the integer in `path` (created on first use): the shared chunk counter."""
    fd = os.open(path, os.O_RDWR | os.O_CREAT, 0o644)
    try:
        fcntl.flock(fd, fcntl.LOCK_EX)
        raw = os.read(fd, 32)
        k = int(raw) if raw.strip() else 0
        os.lseek(fd, 0, os.SEEK_SET)
        os.ftruncate(fd, 0)
        os.write(fd, str(k + 1).encode())
        return k
    finally:
        fcntl.flock(fd, fcntl.LOCK_UN)
        os.close(fd)


def queue_path(train_dir):
    """The shared patient counter of THIS launch: keyed on the launcher (its pid is the parent of every rank), the rendezvous port,
    the elastic run id and the restart count -- workers restarted by torchrun (--max-restarts) get a fresh counter, not the
    exhausted one of their previous life."""
    key = "_".join(str(os.environ.get(k, "0")) for k in ("TORCHELASTIC_RUN_ID", "MASTER_PORT", "TORCHELASTIC_RESTART_COUNT"))
    key = "".join(ch if ch.isalnum() or ch in "_-" else "-" for ch in key)
    return os.path.join(train_dir, f".patient_queue_{key}_{os.getppid()}")


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", required=True)
    ap.add_argument("--pan-list", required=True)
    ap.add_argument("--exe", default=DEFAULT_EXE)
    ap.add_argument("--exe-args", default="", help="extra arguments for the trainer, e.g. '--pin-route'")
    ap.add_argument("--gather", action="store_true")
    ap.add_argument("--backend", default=None, help="nccl (default with GPUs) or gloo")
    ap.add_argument("--max-batch", type=int, default=1024)
    ap.add_argument("--schedule", choices=("dynamic", "static"), default="dynamic")
    ap.add_argument("--resident", type=int, default=1024, help="patients a trainer keeps on its GPU at once (continuous admission)")
    ap.add_argument("--chunk", type=int, default=0, help="ignored (round 4's chunked queue); kept so that old command lines still parse")
    ap.add_argument("--timeout-hours", type=float, default=48.0,
                    help="process-group timeout: how long a finished rank waits for the slowest one")
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
    dist = None

    cfg = json.load(open(args.cfg))
    pans = [ln.split()[0] for ln in open(args.pan_list) if ln.strip()]     # (a second column, if any, is not trusted: counted below)
    ns = [count_observations(cfg, p) for p in pans]
    extra = args.exe_args.split()

    def run_trainer(idx, tag, more=()):
        """One trainer process on the patients `idx` (list file: PAN and observation count per line); returns (exit status, the
        patients it finished -- with a shared queue that is only known afterwards)."""
        shard_file = os.path.join(cfg["exp_train_dir"], f"pan_{tag}.txt")
        with open(shard_file, "w") as f:
            f.write("".join(f"{pans[i]} {ns[i]}\n" for i in idx))
        r = subprocess.run([args.exe, "--cfg", args.cfg, "--pan-list", shard_file, "--device", str(device),
                            "--max-batch", str(args.max_batch), "--resident", str(args.resident)] + list(more) + extra,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        with open(os.path.join(cfg["exp_train_dir"], f"train_{tag}.log"), "w") as f:
            f.write(r.stdout)
        index = {p: i for i, p in enumerate(pans)}
        done = [index[m] for m in re.findall(r"^finish individual id: (\S+) w/", r.stdout, flags=re.M) if m in index]
        return shard.exit_status(r.returncode), done

    rc = 0
    mine = []          # global indices of the patients this rank trained
    t_busy = 0.0
    qfile = None
    if args.schedule == "static" or world == 1:
        parts = shard.lpt_partition(ns, world, Q=int(cfg["Q"]))
        part = [int(i) for i in parts[rank]]
        if part:
            t0 = time.perf_counter()
            rc, mine = run_trainer(part, f"shard_rank{rank}")
            t_busy = time.perf_counter() - t0
    else:
        if int(os.environ.get("LOCAL_WORLD_SIZE", world)) != world:
            sys.exit("train_cohort: --schedule dynamic shares one counter file between the ranks of ONE node; use --schedule static across nodes")
        qfile = queue_path(cfg["exp_train_dir"])
        t0 = time.perf_counter()
        rc, mine = run_trainer(list(range(len(pans))), f"queue_rank{rank}", ["--queue", qfile, "--share", str(world)])
        t_busy = time.perf_counter() - t0
    mine = sorted(set(mine))
    with open(os.path.join(cfg["exp_train_dir"], f"train_rank{rank}.busy"), "w") as f:
        f.write(f"{t_busy:.6f} {len(mine)}\n")
    # MEDGP_FORCE_COLLECTIVES=1: also with ONE rank (under torch.distributed.run): lets a one-GPU box run the RCCL code path of the
    # multi-GPU launch -- process group, all-reduce of the exit status, all-gather, barrier (tests/test_cohort_launchers_gpu.py)
    if world > 1 or (os.environ.get("MEDGP_FORCE_COLLECTIVES") == "1" and "MASTER_ADDR" in os.environ):
        # The process group is created only now, AFTER the training subprocess has returned: lock-step SCG runs until the
        # slowest patient of a shard finishes, so ranks can arrive hours apart, and a group created up front would have its
        # first collective (below) aborted by the default watchdog timeout (nccl 10 min).  The generous timeout covers the
        # rendezvous skew itself.
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
        if args.gather and rc == 0:
            H = int(cfg["D"]) + int(cfg["Q"]) * (int(cfg["D"]) * int(cfg["R"]) + 2 + int(cfg["D"])) if int(cfg["kernel_index"]) == 7 \
                else (1 + 3 * int(cfg["Q"]) if int(cfg["kernel_index"]) == 8 else 3)
            # fixed-size contribution per rank: rows of (global patient index, flag, theta); absent rows are NaN
            cnt = torch.tensor([len(mine)], dtype=torch.int64, device=dev)
            dist.all_reduce(cnt, op=dist.ReduceOp.MAX)
            rows = max(int(cnt.item()), 1)
            buf = torch.full((rows, H + 2), float("nan"), dtype=torch.float64, device=dev)
            for k, gi in enumerate(mine):
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
    if rank == 0 and qfile:     # (after the barrier: every trainer has seen the exhausted counter)
        try:
            os.remove(qfile)
        except OSError:
            pass
    return rc


if __name__ == "__main__":
    sys.exit(main())
