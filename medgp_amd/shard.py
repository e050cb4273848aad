"""Cohort sharding across the GPUs of one node.

Patients are independent (no cross-patient term anywhere in c_inference_exact.cpp /
c_inference_prior.cpp; the reference fans patients out as separate scheduler jobs,
ref: medgpc/util/run_exp_generator.py:213-260), so the cohort is partitioned with no data-path
collective: static LPT (longest processing time first) by the per-patient cost N^3 + c Q N^2.
"""
import numpy as np


def cost(n, Q=5, c=80.0):
    n = np.asarray(n, dtype=np.float64)
    return n ** 3 + c * Q * n * (n + 1) / 2 + 6 * n * n


def lpt_partition(ns, world_size, Q=5):
    """Returns a list (len world_size) of index arrays; deterministic (ties broken by index)."""
    ns = np.asarray(ns)
    order = np.lexsort((np.arange(ns.shape[0]), -cost(ns, Q)))
    loads = np.zeros(world_size)
    parts = [[] for _ in range(world_size)]
    for i in order:
        r = int(np.argmin(loads))   # argmin returns the first minimum: deterministic
        parts[r].append(int(i))
        loads[r] += cost(ns[i], Q)
    return [np.array(sorted(p), dtype=np.int64) for p in parts]


def weak_shard(P_per_rank, rank):
    """Weak scaling: every rank owns P_per_rank patients; global patient ids are contiguous."""
    return np.arange(rank * P_per_rank, (rank + 1) * P_per_rank, dtype=np.int64)


def exit_status(returncode):
    """Exit status of a child as a NON-NEGATIVE int that survives max(): subprocess reports a child killed by signal s as -s
    (a HIP fault -> SIGABRT = -6, the OOM killer = -9), which would lose against the 0 of the healthy ranks in
    all_reduce(MAX) and in `rc = max(rc, ...)`.  The shell convention: 128 + s."""
    return returncode if returncode >= 0 else 128 - returncode
