"""Worker for test_rccl_branches_with_one_rank (launched by torch.distributed.run with ONE rank on the GPU box): the collective path of
cohort_mode.kde_modes under the nccl (= RCCL) backend with the real HIP kernels, forced by MEDGP_FORCE_COLLECTIVES=1."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from medgp_amd import cohort_mode  # noqa: E402


def main():
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    rng = np.random.default_rng(3)
    series = [rng.normal(size=n) for n in (50, 400, 70, 20, 190, 330, 30)]
    os.environ.pop("MEDGP_FORCE_COLLECTIVES", None)
    direct = cohort_mode.kde_modes(series, True)                 # one rank: straight to the kernel
    os.environ["MEDGP_FORCE_COLLECTIVES"] = "1"
    coll = cohort_mode.kde_modes(series, True)                   # the same through all_reduce + all_gather on device tensors
    assert dist.get_backend() == "nccl" and np.array_equal(direct, coll), (direct, coll)
    dist.barrier()
    print("RCCL_COHORT_MODE_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
