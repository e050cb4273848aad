"""one batched medgp_kde_mode call for rocprofv3: 326 series x 4096 samples (one cluster of a 4096-subject cohort at D = 24)"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medgp_amd import capi
rng = np.random.default_rng(2024)
series = [rng.normal(size=4096) * rng.uniform(0.1, 3.0) for _ in range(326)]
capi.kde_mode(series[:8], True, 0)
for _ in range(3):
    _, _, st, kms = capi.kde_mode(series, True, 0, full=True)
print("kernel ms", kms)
