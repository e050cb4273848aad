"""Synthetic experiments on disk (tests, bench.py's host legs, scratch timings).
Builds a MedGP experiment directory in the reference's on-disk formats:
exp_setup.json (keys of medgpc/util/config.py:5-35), hyp_bound.txt (config.py:38-66), per-patient
feature<idx>.txt (count, then t / v pairs; README.md:64-72) and feature<idx>_stat.bin (mean, std)."""
import json
import os

import numpy as np

from medgp_amd import synth

OPT = dict(random_init_num=6, random_seed=718, top_iteration_num=3, iteration_num_per_update=8,
           online_learn_rate=0.00001, online_momentum=0.9,
           lower_bound_noise=0.15, upper_bound_noise=0.4, lower_bound_a=-1.5, upper_bound_a=1.5,
           lower_bound_period=12, upper_bound_period=72, lower_bound_lengthscale=6, upper_bound_lengthscale=72,
           lower_bound_lambda=0.1, upper_bound_lambda=0.5, lower_bound_scale=0.5, upper_bound_scale=2.0)


def make_experiment(root, pans, D=2, Q=3, R=2, N=60, prior_index=2, feature_index=(18, 19), seed=5, opt=None, kernel_index=7):
    """kernel_index 7: LMC-SM (multi-output); 0: SE, 8: SM -- the single-output families (D = 1, one feature; prior_index 0),
    bounds in the order medgpc/util/config.py:67-100 writes them."""
    opt = dict(OPT, **(opt or {}))
    if kernel_index != 7:
        D, R, prior_index = 1, 1, 0
        Q = 1 if kernel_index == 0 else Q
    root = str(root)
    dirs = {k: os.path.join(root, k) for k in ("data", "train", "test", "kernel", "cfg")}
    for d in dirs.values():
        os.makedirs(d, exist_ok=True)
    feature_index = list(feature_index)[:D]
    stats = [(37.0 + 3 * j, 4.0 + j) for j in range(D)]
    for j, fi in enumerate(feature_index):
        np.array(stats[j], dtype=np.float64).tofile(os.path.join(dirs["data"], f"feature{fi}_stat.bin"))
    raw = {}
    for p, pan in enumerate(pans):
        os.makedirs(os.path.join(dirs["data"], pan), exist_ok=True)
        m, t, y = synth.patient(seed, p, D, N if isinstance(N, int) else N[p])
        raw[pan] = {}
        for j, fi in enumerate(feature_index):
            tt, vv = t[m == j], (y[m == j].astype(np.float64) * stats[j][1] + stats[j][0]).astype(np.float32)
            raw[pan][j] = (tt, vv)
            with open(os.path.join(dirs["data"], pan, f"feature{fi}.txt"), "w") as f:
                f.write(f"{len(tt)}\n")
                for a, b in zip(tt, vv):
                    f.write(f"{a:.6f}\n{b:.6f}\n")
    cfg = {"data_dir": dirs["data"], "exp_top_dir": root, "exp_train_dir": dirs["train"], "exp_test_dir": dirs["test"],
           "exp_kernel_dir": dirs["kernel"], "exp_cfg_dir": dirs["cfg"], "hyp_bound_file": "hyp_bound.txt",
           "kernel": {7: "LMC-SM", 0: "SE", 8: "SM"}[kernel_index], "kernel_index": kernel_index,
           "prior": "hier-gamma" if prior_index == 2 else "none", "prior_index": prior_index,
           "Q": Q, "D": D, "R": R, "feature_index": "".join(f"{f} " for f in feature_index),
           "eta": 0.01, "beta_lam": 0.01, "cv_fold_num": 1, "cv_assign_file": "cv_assign.txt",
           "random_init_num": opt["random_init_num"], "random_seed": opt["random_seed"],
           "top_iteration_num": opt["top_iteration_num"], "iteration_num_per_update": opt["iteration_num_per_update"],
           "online_learn_rate": opt["online_learn_rate"], "online_momentum": opt["online_momentum"]}
    cfg_file = os.path.join(dirs["cfg"], "exp_setup.json")
    json.dump(cfg, open(cfg_file, "w"), indent=4)
    with open(os.path.join(dirs["cfg"], "hyp_bound.txt"), "w") as f:
        def w(lo, hi, n):
            for _ in range(n):
                f.write("{:6.6f}\n".format(lo))
                f.write("{:6.6f}\n".format(hi))
        if kernel_index == 7:
            w(opt["lower_bound_noise"], opt["upper_bound_noise"], D)
            w(opt["lower_bound_a"], opt["upper_bound_a"], Q * D * R)
            w(opt["lower_bound_period"], opt["upper_bound_period"], Q)
            w(opt["lower_bound_lengthscale"], opt["upper_bound_lengthscale"], Q)
            w(opt["lower_bound_lambda"], opt["upper_bound_lambda"], Q * D)
        elif kernel_index == 0:      # ref config.py:67-79: noise, length scale, scale factor
            w(opt["lower_bound_noise"], opt["upper_bound_noise"], 1)
            w(opt["lower_bound_lengthscale"], opt["upper_bound_lengthscale"], 1)
            w(opt["lower_bound_scale"], opt["upper_bound_scale"], 1)
        else:                        # ref config.py:81-100: noise, Q scale factors, Q periods, Q length scales
            w(opt["lower_bound_noise"], opt["upper_bound_noise"], 1)
            w(opt["lower_bound_scale"], opt["upper_bound_scale"], Q)
            w(opt["lower_bound_period"], opt["upper_bound_period"], Q)
            w(opt["lower_bound_lengthscale"], opt["upper_bound_lengthscale"], Q)
    return {"cfg": cfg_file, "dirs": dirs, "stats": stats, "raw": raw, "opt": opt, "Q": Q, "D": D, "R": R,
            "feature_index": feature_index}


CONFIG1_REL = os.path.join("tests", "golden", "ref_cfg", "PT_INR")
CONFIG1_STATS = ((37.0, 4.0), (1.3, 0.4))     # (mean, std) of features 18 (PT) and 19 (INR): feature<idx>_stat.bin


def reference_config1_tree(tmp, repo_root, pan="PT0001", n=150, seed=41):
    """BASELINE config 1 on the REFERENCE-WRITTEN configuration (tests and bench.py's config-1 leg): copies tests/golden/ref_cfg/PT_INR
    (exp_setup.json + hyp_bound.txt written by the reference's config.py, kernel/fold0/gmm_mode_* by its binaryIO.py) under `tmp`
    KEEPING the relative paths the JSON names, and writes one synthetic patient with n / 2 + n / 2 observations next to it as
    feature18.txt / feature19.txt + feature<idx>_stat.bin (count, then t / value pairs with 6 decimals; ref README.md:64-72).  The
    hosts are then run with cwd = tmp and --cfg <returned relative path>, as the reference's CLI is run from its repository root.
    Returns (cwd, cfg path relative to it, meta, t, y) with t / y as the loader reads them back (text round trip, z-scored in
    double, narrowed to float: ref dataio/c_experiment.cpp:296-305)."""
    import shutil
    D = 2
    dst = os.path.join(str(tmp), CONFIG1_REL)
    shutil.copytree(os.path.join(repo_root, CONFIG1_REL), dst)
    for sub in ("train", "test", "data", os.path.join("data", pan)):
        os.makedirs(os.path.join(dst, sub), exist_ok=True)
    m, t, y = synth.patient(seed, 0, D, n)
    tl, yl = [], []
    for j, fi in enumerate((18, 19)):
        np.array(CONFIG1_STATS[j], np.float64).tofile(os.path.join(dst, "data", f"feature{fi}_stat.bin"))
        tt = t[m == j]
        vv = (y[m == j].astype(np.float64) * CONFIG1_STATS[j][1] + CONFIG1_STATS[j][0]).astype(np.float32)
        with open(os.path.join(dst, "data", pan, f"feature{fi}.txt"), "w") as f:
            f.write(f"{len(tt)}\n")
            for a, b in zip(tt, vv):
                f.write(f"{a:.6f}\n{b:.6f}\n")
        t6 = np.array([np.float32(f"{a:.6f}") for a in tt], np.float32)
        v6 = np.array([np.float32(f"{b:.6f}") for b in vv], np.float32)
        tl.append(t6)
        yl.append(((v6.astype(np.float64) - CONFIG1_STATS[j][0]) / CONFIG1_STATS[j][1]).astype(np.float32))
    return str(tmp), os.path.join(CONFIG1_REL, "exp_setup.json"), m, np.concatenate(tl), np.concatenate(yl)
