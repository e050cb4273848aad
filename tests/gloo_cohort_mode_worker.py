"""Worker for test_cohort_mode_two_ranks_gloo (launched by torch.distributed.run): the N>1 path of cohort_mode on CPU,
with the oracle standing in for the HIP kernel (test infrastructure; the product path has no CPU evaluator)."""
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from medgp_amd import cohort_mode  # noqa: E402
from oracle import kde_oracle as KO  # noqa: E402
from test_cohort_mode import make_cohort, oracle_fn  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    c = make_cohort(5, P=23, Q=3, D=3, R=2, newQ=2)
    out_dir = sys.argv[1]
    exp = dict(c["exp"], exp_kernel_dir=out_dir)
    got = cohort_mode.output_mode_LMC_SM(-1, exp, c["pan"], c["hyp"], c["mpan"], c["midx"], 2, c["assign"], "kmeans", kde_fn=oracle_fn)
    want = KO.output_mode_lmc_sm(3, 3, 2, c["pan"], c["hyp"], c["mpan"], c["midx"], 2, c["assign"])
    assert np.array_equal(got, want), np.abs(got - want).max()
    # ragged series, more ranks than some shards need
    rng = np.random.default_rng(3)
    series = [rng.normal(size=n) for n in (5, 40, 7, 2, 19, 33, 3)]
    m = cohort_mode.kde_modes(series, True, oracle_fn)
    assert np.array_equal(m, oracle_fn(series, True))
    grids = [None, np.linspace(-3, 3, 50), None, np.linspace(-1, 1, 7), None, None, np.linspace(0, 2, 11)]
    m = cohort_mode.kde_modes(series, False, oracle_fn, tests=grids)
    assert np.array_equal(m, oracle_fn(series, False, grids))
    # a series only ONE rank owns fails (the reference's KDE fit raises on it): every rank must raise -- nobody may be left
    # waiting in the all_gather (round-2 advisor finding)
    from medgp_amd import capi

    def failing_fn(ss, w, tt=None):
        if any(len(x) < 2 for x in ss):
            raise capi.MedgpError("KDE fit failed for a series with fewer than two samples")
        return oracle_fn(ss, w, tt)
    bad = [rng.normal(size=n) for n in (30, 25, 1, 28)]            # the 1-sample series is dealt to exactly one rank
    owner = cohort_mode.deal_series([len(x) * 2 * len(x) for x in bad], dist.get_world_size())
    raised = False
    try:
        cohort_mode.kde_modes(bad, True, failing_fn)
    except capi.MedgpError as e:
        raised = True
        mine = owner[2] == rank
        assert ("fewer than two" in str(e)) == bool(mine), (rank, str(e))
    assert raised
    # and the group is still usable afterwards
    m = cohort_mode.kde_modes(series, True, oracle_fn)
    assert np.array_equal(m, oracle_fn(series, True))
    dist.barrier()
    if rank == 0:
        assert os.path.exists(os.path.join(out_dir, "all", "kmeans_mode_param.bin"))
        print("GLOO_COHORT_MODE_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
