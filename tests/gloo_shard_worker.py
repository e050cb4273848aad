"""Worker for test_two_rank_gloo_sharded_evaluation_matches_single_process (launched by torch.distributed.run)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from medgp_amd import shard, synth  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    D, N, Q, R, P = 2, 24, 2, 2, 6
    ns = [N + 3 * p for p in range(P)]
    parts = shard.lpt_partition(ns, world)
    mine = parts[rank]
    res = np.zeros((P, 1 + synth.num_hyp(7, Q, D, R)))
    for p in mine:
        m, t, y = synth.patient(21, int(p), D, ns[p])
        r = O.nlml_grad(7, Q, D, R, m, t, y, synth.theta(21, int(p), 7, Q, D, R))
        res[p, 0], res[p, 1:] = r["nlml"], r["grad"]
    tt = torch.from_numpy(res)
    dist.all_reduce(tt)   # disjoint rows: the sum is a gather
    # bench.py's timing aggregation: max over ranks, whole-job units
    agg = bench.aggregate_time(0.5 + rank, world)
    assert abs(agg - (0.5 + world - 1)) < 1e-12
    if rank == 0:
        full = np.zeros_like(res)
        for p in range(P):
            m, t, y = synth.patient(21, p, D, ns[p])
            r = O.nlml_grad(7, Q, D, R, m, t, y, synth.theta(21, p, 7, Q, D, R))
            full[p, 0], full[p, 1:] = r["nlml"], r["grad"]
        assert np.array_equal(tt.numpy(), full)
        line = bench.result_line(value=123.0, n_gpus=world, steps=2, warmup=1, ms_per_step=3.0, workload="t", extra={})
        assert line["n_gpus"] == world and line["scaling"] == "weak" and line["vs_baseline"] is None
        print("GLOO_SHARD_OK")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
