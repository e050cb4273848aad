"""Cohort mode kernel: the step after per-patient training.

Mirrors medgpc/clustering/mode_estimate.py (same function names, arguments, output files):
    output_mode_kernel(...)   ref: mode_estimate.py:8-27     dispatch on exp_param["kernel"]
    output_mode_LMC_SM(...)   ref: mode_estimate.py:242-435  nuggets, per-cluster mu / v, element-wise mode of the aggregated
                                                             B matrices, SVD back to (A, lambda), the two output files
    output_mode_SE(...)       ref: mode_estimate.py:29-79    arg-max modes; the length-scale density on a 100001-point grid
    output_mode_SM(...)       ref: mode_estimate.py:82-240   arg-max modes; mu / v on reciprocal period / length-scale grids
The D + newQ (2 + D(D+1)/2) kernel density estimates (each O(P^2) Gaussian terms over the cohort) are ONE batched call of
the HIP kernel behind medgp_kde_mode; there is no CPU path (the library raises without a GPU).
Multi-GPU (one process per GPU, torch.distributed): the series are dealt to the ranks by cost (longest first), every rank
runs its share on its own GPU, one all_gather brings the modes together -- the only step of the whole system where a
collective carries data (RCCL over xGMI when the backend is nccl).  The trained hypers it starts from are what
`python -m medgp_amd.train_cohort --gather` leaves in cohort_train_hyp.npy (an all-gather over the same ranks).
Figures (plotting_mode != 0) are the reference's matplotlib side and are not produced.
"""
import os
from array import array

import numpy as np

from . import capi


def _dist():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist
    except Exception:   # pragma: no cover
        pass
    return None


def deal_series(costs, world):
    """Longest-processing-time assignment of series to ranks: returns owner[len(costs)].  Deterministic (ties by index)."""
    order = sorted(range(len(costs)), key=lambda i: (-float(costs[i]), i))
    load = [0.0] * world
    owner = np.zeros(len(costs), dtype=np.int64)
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        owner[i] = r
        load[r] += float(costs[i])
    return owner


def _all_gather_padded(vec, group=None):
    """all_gather of one float64 vector per rank (lengths may differ): list of numpy arrays, one per rank."""
    import torch
    dist = _dist()
    world = dist.get_world_size(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    n = torch.tensor([len(vec)], dtype=torch.int64, device=dev)
    ns = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(ns, n, group=group)
    ns = [int(v.item()) for v in ns]
    buf = torch.zeros(max(max(ns), 1), dtype=torch.float64, device=dev)
    buf[:len(vec)] = torch.as_tensor(np.asarray(vec, dtype=np.float64), device=dev)
    bufs = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf, group=group)
    return [b[:k].cpu().numpy() for b, k in zip(bufs, ns)]


def kde_modes(series, weighted=True, kde_fn=None, device=None, group=None, tests=None):
    """Modes of all series; sharded over the ranks of `group` when torch.distributed is initialised.
    tests: optional list of evaluation grids (None entries = at the samples).
    kde_fn(list_of_arrays, weighted, list_of_grids_or_None) -> modes defaults to the HIP kernels on this rank's GPU."""
    if kde_fn is None:
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        kde_fn = lambda ss, w, tt: _device_modes(ss, w, device, tt)   # noqa: E731
    dist = _dist()
    world = dist.get_world_size(group) if dist else 1
    if world == 1 and not (dist and os.environ.get("MEDGP_FORCE_COLLECTIVES") == "1"):   # (forced: the collective path with one rank, for tests of the RCCL branch)
        return np.asarray(kde_fn(series, weighted, tests), dtype=np.float64)
    rank = dist.get_rank(group)
    cost = [len(s) * (len(s) + (len(tests[i]) if tests is not None and tests[i] is not None else len(s))) for i, s in enumerate(series)]
    owner = deal_series(cost, world)
    mine = np.where(owner == rank)[0]
    sub_t = None if tests is None else [tests[i] for i in mine]
    # A failing series (KDEUnivariate.fit raises; the reference exits the whole program there, ref: mode_estimate.py:23-26) is
    # local to the rank that owns it.  Every rank must still take part in the SAME sequence of collectives: catch the local
    # error, agree on failure with one all_reduce(MAX), then raise on EVERY rank -- otherwise the healthy ranks would sit in
    # the all_gather until the process group times out.
    err, local = None, np.zeros(0)
    try:
        if len(mine):
            local = np.asarray(kde_fn([series[i] for i in mine], weighted, sub_t), dtype=np.float64)
    except Exception as e:   # noqa: BLE001 -- any local failure must reach the collective below
        err = e
    import torch
    flag = torch.tensor([1 if err is not None else 0], dtype=torch.int32)
    if dist.get_backend(group) == "nccl":
        flag = flag.cuda()
    dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
    if int(flag.item()):
        if err is not None:
            raise err
        raise capi.MedgpError("KDE fit failed on another rank (see that rank's message)")
    parts = _all_gather_padded(local, group)
    out = np.full(len(series), np.nan)
    for r in range(world):
        out[np.where(owner == r)[0]] = parts[r]
    return out


def _device_modes(series, weighted, device, tests=None):
    mode, _, st, _ = capi.kde_mode(series, weighted, device, full=True, test=tests)
    if np.any(st < 0):
        # KDEUnivariate.fit raises on these; output_mode_kernel's handler then exits (ref: mode_estimate.py:23-26)
        raise capi.MedgpError(f"KDE fit failed for series {np.where(st < 0)[0].tolist()} (fewer than two samples, a non-finite "
                              "sample, or zero bandwidth)")
    return mode


def write_double_to_bin(filename, d_array):
    """ref: medgpc/util/binaryIO.py:6-10 (native doubles, no header)."""
    with open(filename, "wb") as f:
        array("d", np.asarray(d_array, dtype=np.float64).ravel().tolist()).tofile(f)


def output_mode_kernel(fold, exp_param, pan_array, hyp_array, mixture_pan, mixture_index, mixture_cluster_num,
                       mixture_cluster_assign, kernclust_alg, plotting_mode=0, plotting_param=None, **kw):
    """ref: mode_estimate.py:8-27."""
    by_kernel = {"SE": output_mode_SE, "SM": output_mode_SM, "LMC-SM": output_mode_LMC_SM}
    if exp_param["kernel"] not in by_kernel:
        print("Error: specified kernel type {} not supported".format(exp_param["kernel"]))
        raise NotImplementedError
    return by_kernel[exp_param["kernel"]](fold, exp_param, pan_array, hyp_array, mixture_pan, mixture_index, mixture_cluster_num,
                                          mixture_cluster_assign, kernclust_alg, plotting_mode, plotting_param, **kw)


def _write_mode_files(exp_param, fold, kernclust_alg, newQ, kde_mode_hyp, write, group):
    """ref: mode_estimate.py:69-73 / :227-231 / :427-431 -- the two files main_one_test reads (c_experiment.cpp:179-219)."""
    dist = _dist()
    if not write or (dist is not None and dist.get_rank(group) != 0):
        return
    sub_dir = "fold{}".format(fold) if fold != -1 else "all"
    kde_output_dir = os.path.join(exp_param["exp_kernel_dir"], sub_dir)
    os.makedirs(kde_output_dir, exist_ok=True)
    prefix = "mode_"
    np.savetxt(os.path.join(kde_output_dir, "{}_{}mixture_num.txt".format(kernclust_alg, prefix)), [newQ], fmt="%d")
    mode_file_name = os.path.join(kde_output_dir, "{}_{}param.bin".format(kernclust_alg, prefix))
    write_double_to_bin(mode_file_name, np.asarray(kde_mode_hyp).flatten())
    print("Info: output final mode parameters to file: {}".format(mode_file_name))


_GRID = (0.01, 1000.0, 100001)   # np.linspace arguments of the reference's length-scale / period grids


def output_mode_SE(fold, exp_param, pan_array, hyp_array, mixture_pan, mixture_index, mixture_cluster_num,
                   mixture_cluster_assign, kernclust_alg, plotting_mode=0, plotting_param=None, kde_fn=None, device=None,
                   group=None, write=True):
    """ref: mode_estimate.py:29-79.  Hypers (nugget, lengthscale, scalefactor): arg-max modes, the length-scale on a grid."""
    assert mixture_cluster_num == 1                                                # ref :44
    hyp_array = np.asarray(hyp_array, dtype=np.float64)
    assert len(hyp_array) == len(pan_array)
    series = [np.exp(hyp_array[:, i]) for i in range(hyp_array.shape[1])]
    tests = [np.linspace(*_GRID) if i == 1 else None for i in range(hyp_array.shape[1])]   # ref :52-60
    kde_mode_hyp = np.log(kde_modes(series, False, kde_fn, device, group, tests))
    _write_mode_files(exp_param, fold, kernclust_alg, mixture_cluster_num, kde_mode_hyp, write, group)
    return kde_mode_hyp


def output_mode_SM(fold, exp_param, pan_array, hyp_array, mixture_pan, mixture_index, mixture_cluster_num,
                   mixture_cluster_assign, kernclust_alg, plotting_mode=0, plotting_param=None, kde_fn=None, device=None,
                   group=None, write=True):
    """ref: mode_estimate.py:82-240.  D = 1; hypers (nugget, w_q, mu_q, sqrt v_q); all modes are arg-max modes, mu and v on the
    reciprocal period / length-scale grids (ref: :170-184), the weights of a subject's components in one cluster are added."""
    Q = exp_param["Q"]
    assert exp_param["D"] == 1                                                     # ref :99
    newQ = int(mixture_cluster_num)
    pan_array = np.asarray(pan_array)
    hyp_array = np.asarray(hyp_array, dtype=np.float64)
    mixture_pan, mixture_index = np.asarray(mixture_pan), np.asarray(mixture_index)
    mixture_cluster_assign = np.asarray(mixture_cluster_assign)
    assert len(hyp_array) == len(pan_array)
    cluster_id_array = np.unique(mixture_cluster_assign)
    assert len(cluster_id_array) == newQ
    row_of = {p: i for i, p in enumerate(pan_array.tolist())}
    per = np.linspace(*_GRID)
    series, tests = [np.exp(hyp_array[:, 0])], [None]                              # nugget, ref :108-112
    for cid in cluster_id_array:
        comp = np.where(mixture_cluster_assign == cid)[0]
        assert len(comp) > 0
        rows = np.array([row_of[p] for p in mixture_pan[comp].tolist()])
        qq = mixture_index[comp].astype(np.int64)
        series.append(np.exp(hyp_array[rows, 1 + Q + qq])); tests.append(1.0 / per)                       # ref :170-173
        series.append(np.exp(hyp_array[rows, 1 + 2 * Q + qq])); tests.append(1.0 / (2.0 * np.pi * per))   # ref :181-184
        cpan = mixture_pan[comp]
        w = [sum(np.exp(hyp_array[row_of[pan], 1 + q1]) for q1 in mixture_index[comp][cpan == pan]) for pan in np.unique(cpan).tolist()]
        series.append(np.asarray(w)); tests.append(None)                           # ref :198-220
    modes = kde_modes(series, False, kde_fn, device, group, tests)
    kde_mode_hyp = np.zeros(1 + 3 * newQ)
    kde_mode_hyp[0] = np.log(modes[0])
    for q in range(newQ):
        kde_mode_hyp[1 + newQ + q] = np.log(modes[1 + 3 * q])
        kde_mode_hyp[1 + 2 * newQ + q] = np.log(modes[2 + 3 * q])
        kde_mode_hyp[1 + q] = np.log(modes[3 + 3 * q])
    _write_mode_files(exp_param, fold, kernclust_alg, newQ, kde_mode_hyp, write, group)
    return kde_mode_hyp


def output_mode_LMC_SM(fold, exp_param, pan_array, hyp_array, mixture_pan, mixture_index, mixture_cluster_num,
                       mixture_cluster_assign, kernclust_alg, plotting_mode=0, plotting_param=None, kde_fn=None,
                       device=None, group=None, write=True):
    """ref: mode_estimate.py:242-435.  Returns kde_mode_hyp; rank 0 writes
    <exp_kernel_dir>/<fold dir>/<alg>_mode_mixture_num.txt and _mode_param.bin (ref: :427-431)."""
    Q, D, R = exp_param["Q"], exp_param["D"], exp_param["R"]
    newQ = int(mixture_cluster_num)
    pan_array = np.asarray(pan_array)
    hyp_array = np.asarray(hyp_array, dtype=np.float64)
    mixture_pan, mixture_index = np.asarray(mixture_pan), np.asarray(mixture_index)
    mixture_cluster_assign = np.asarray(mixture_cluster_assign)
    assert hyp_array.shape == (len(pan_array), D + Q * (D * R + 2 + D))
    cluster_id_array = np.unique(mixture_cluster_assign)                          # ref :287-289
    assert len(cluster_id_array) == newQ
    row_of = {p: i for i, p in enumerate(pan_array.tolist())}
    assert len(row_of) == len(pan_array)                                          # ref :299-300 (one row per id)

    series = [np.exp(hyp_array[:, d]) for d in range(D)]                           # nuggets, ref :273-276
    tri = [(d1, d2) for d1 in range(D) for d2 in range(d1, D)]
    for cid in cluster_id_array:                                                   # ref :318-413
        comp = np.where(mixture_cluster_assign == cid)[0]
        assert len(comp) > 0
        rows = np.array([row_of[p] for p in mixture_pan[comp].tolist()])
        qq = mixture_index[comp].astype(np.int64)
        series.append(np.exp(hyp_array[rows, D + Q * D * R + qq]))                 # mu, ref :326-337
        series.append(np.exp(hyp_array[rows, D + Q * D * R + Q + qq]))             # sqrt v
        # aggregated B of every subject with components in this cluster, ref :369-386 (np.unique order)
        cpan = mixture_pan[comp]
        upan = np.unique(cpan)
        all_B = np.zeros((len(upan), D, D))
        for k, pan in enumerate(upan.tolist()):
            hyp = hyp_array[row_of[pan]]
            for q1 in mixture_index[comp][cpan == pan]:
                A = hyp[D + q1 * D * R: D + (q1 + 1) * D * R].reshape(D, R)
                lam = np.exp(hyp[D + Q * (D * R + 2) + q1 * D: D + Q * (D * R + 2) + (q1 + 1) * D])
                all_B[k] += A @ A.T + np.diag(lam)
        for d1, d2 in tri:                                                         # ref :406-409
            series.append(np.ascontiguousarray(all_B[:, d1, d2]))

    modes = kde_modes(series, True, kde_fn, device, group)                         # compute_kde + compute_mode(weighted=True)

    kde_mode_hyp = np.zeros(D + newQ * (D * R + 2 + D))
    kde_mode_hyp[:D] = np.log(modes[:D])                                           # ref :279
    pos = D
    for q in range(newQ):
        kde_mode_hyp[D + newQ * D * R + q] = np.log(modes[pos])                    # ref :342
        kde_mode_hyp[D + newQ * (D * R + 1) + q] = np.log(modes[pos + 1])          # ref :353
        pos += 2
        kde_B = np.zeros((D, D))
        for d1, d2 in tri:                                                         # ref :410-413
            kde_B[d1, d2] = kde_B[d2, d1] = modes[pos]
            pos += 1
        U, S, _ = np.linalg.svd(kde_B)                                             # ref :423-431
        A_ = (U * np.sqrt(S))[:, 0:R]
        lam_ = np.diag(kde_B - np.dot(A_, A_.T)).copy()
        lam_[np.where(lam_ <= 0.0)] = 1e-15
        kde_mode_hyp[D + newQ * (D * R + 2) + q * D: D + newQ * (D * R + 2) + (q + 1) * D] = np.log(lam_)
        kde_mode_hyp[D + q * D * R: D + (q + 1) * D * R] = A_.reshape(-1)

    _write_mode_files(exp_param, fold, kernclust_alg, newQ, kde_mode_hyp, write, group)
    return kde_mode_hyp
