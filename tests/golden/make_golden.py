#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/ (run in the BUILD container only;
it reads /root/reference, which does not exist on the GPU box).

1. fastkernel_*.npz  -- outputs of the reference's own Python statement of B_q and k_q
   (/root/reference/medgpc/visualization/fastkernel.py:13-48), imported here.
2. appendixA_*.npz   -- inputs of SURVEY.md Appendix A's driver (gen_appendixA_inputs.cpp, libstdc++
   distributions on mt19937(1234)) together with the nlml the COMPILED REFERENCE printed for them, as
   recorded in SURVEY.md section 8c (the reference cannot be rebuilt under this round's rules: it needs
   <mkl.h> and rapidjson), plus the oracle's own fp64 outputs for regression.
3. ref_prior*.json   -- outputs of the part of the reference's C++ that DOES compile here unmodified (c_prior, c_hyperparam,
   c_inference_prior), see ref_prior() below (round 6).
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

# (D, N, Q, R) -> nlml printed by the compiled reference (SURVEY.md section 8c table, 1 thread)
RECORDED = {
    (2, 150, 5, 2): 807.1832958150,
    (2, 256, 5, 2): 1595.6663842416,
    (24, 512, 5, 8): 1740.5962798340,
    (24, 2048, 5, 8): 10746.7599198414,
}
# prior mode 2 (eta = beta_lam = 0.01), D=24 N=512, 8 threads
RECORDED_PRIOR2 = {(24, 512, 5, 8): 2230.1612403014}


def load_bin(p):
    b = open(p, "rb").read()
    D, N, Q, R, H = [int(v) for v in np.frombuffer(b, np.int32, 5)]
    o = 20
    meta = np.frombuffer(b, np.int32, N, o); o += 4 * N
    x = np.frombuffer(b, np.float32, N, o); o += 4 * N
    y = np.frombuffer(b, np.float32, N, o); o += 4 * N
    th = np.frombuffer(b, np.float64, H, o)
    return D, N, Q, R, meta.copy(), x.copy(), y.copy(), th.copy()


def appendix_a():
    with tempfile.TemporaryDirectory() as td:
        exe = os.path.join(td, "gen")
        subprocess.check_call(["g++", "-O2", "-o", exe, os.path.join(HERE, "gen_appendixA_inputs.cpp")])
        for (D, N, Q, R), ref in RECORDED.items():
            p = os.path.join(td, "a.bin")
            subprocess.check_call([exe, str(D), str(N), str(Q), str(R), "0", p])
            D_, N_, Q_, R_, meta, x, y, th = load_bin(p)
            want_grad = True   # N = 2048 too: 1114 gradient components of the multi-CU path at its real size
            r = O.nlml_grad(7, Q, D, R, meta, x, y, th, flag_grad=want_grad, nthreads=8)
            out = dict(D=D, N=N, Q=Q, R=R, meta=meta, t=x, y=y, theta=th, ref_fp32_nlml=ref,
                       oracle_nlml=r["nlml"], oracle_status=r["status"])
            if want_grad:
                out["oracle_grad"] = r["grad"]
            if (D, N, Q, R) in RECORDED_PRIOR2:
                pr = O.Prior.hier_gamma(Q, D, R, 0.01, 0.01)
                r2 = O.nlml_grad(7, Q, D, R, meta, x, y, th, flag_grad=True, prior=pr, nthreads=8)
                out["ref_fp32_nlml_prior2"] = RECORDED_PRIOR2[(D, N, Q, R)]
                out["oracle_nlml_prior2"] = r2["nlml"]
                out["oracle_grad_prior2"] = r2["grad"]
            np.savez_compressed(os.path.join(HERE, f"appendixA_D{D}_N{N}.npz"), **out)
            print(f"appendixA D={D} N={N}: oracle {r['nlml']:.10f} recorded reference {ref:.10f} "
                  f"rel {abs(r['nlml'] - ref) / ref:.2e}")


def fastkernel():
    # import the single reference file (its package __init__ pulls in seaborn, absent here)
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_fastkernel", "/root/reference/medgpc/visualization/fastkernel.py")
    fk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fk)
    rng = np.random.default_rng(20240601)
    for (Q, D, R) in [(5, 2, 2), (5, 24, 8), (3, 7, 4)]:
        H = D + Q * (D * R + 2 + D)
        hyp = rng.normal(0, 0.7, size=H)
        B = np.stack(fk.compute_B_matrix(Q, D, R, hyp))            # fastkernel.py:13-22
        x = rng.uniform(0, 200, size=(40, 1))
        mu = np.exp(hyp[D + Q * D * R: D + Q * D * R + Q])
        v = np.exp(hyp[D + Q * D * R + Q: D + Q * D * R + 2 * Q])
        # fastkernel's v argument is v^2 (feature_extraction.py:75-77: exp(2 * theta_v))
        resp = np.stack([fk.compute_sm_1d(mu[q], v[q] ** 2, x)[:, 0] for q in range(Q)])   # fastkernel.py:24-48
        np.savez_compressed(os.path.join(HERE, f"fastkernel_Q{Q}_D{D}_R{R}.npz"), Q=Q, D=D, R=R, hyp=hyp, B=B,
                            x=x[:, 0], mu=mu, v=v, resp=resp)
        print(f"fastkernel Q={Q} D={D} R={R}: B {B.shape}, resp {resp.shape}")


def config5():
    """BASELINE config 5 at full size: D=64, N=4096, Q=5, R=8 (H=2954), a random 50 % of the A entries exactly zero
    and clamped (test-time prior, ref: prior/c_prior.cpp:118-140), hierarchical-gamma prior (mode 2, eta = beta_lam =
    0.01; ref: scripts/gen_medgpc_example.sh:11).  Expected values: the oracle (blocked gradient form), fp64."""
    from medgp_amd import synth
    D, N, Q, R, seed = 64, 4096, 5, 8, 5005
    meta, t, y = synth.patient(seed, 0, D, N)
    th = synth.theta(seed, 0, 7, Q, D, R, sparse_frac=0.5)
    pr = O.Prior.hier_gamma(Q, D, R, 0.01, 0.01)
    z = np.where(th[D:D + Q * D * R] == 0.0)[0] + D
    pr.type[z] = 0
    r = O.nlml_grad(7, Q, D, R, meta, t, y, th, flag_grad=True, prior=pr, nthreads=8)
    assert r["ok"] and r["status"] == 0
    np.savez_compressed(os.path.join(HERE, f"config5_D{D}_N{N}.npz"), D=D, N=N, Q=Q, R=R, seed=seed, meta=meta, t=t, y=y,
                        theta=th, clamped=z, oracle_nlml=r["nlml"], oracle_grad=r["grad"], oracle_status=r["status"])
    print(f"config5 D={D} N={N}: oracle nlml {r['nlml']:.10f}, {z.size} clamped A entries")


def fastkernel_gram():
    """Full Gram matrices assembled ONLY from the reference's Python factors: K[i][j] = sum_q B_q[m_i][m_j] k_q(|t_i - t_j|)
    with B_q from fastkernel.compute_B_matrix (fastkernel.py:13-22) and k_q from fastkernel.compute_sm_1d (:33-48) evaluated
    at the pairwise distances (np.pi, as that file uses).  Pins rows a3-a8 (hyper split, B_q, distances, LMC-SM Gram) end to end
    to reference-produced numbers; the noise diagonal sigma_d^2 = exp(2 theta_d) (ref: c_likelihood.cpp:38-43) is added by the test."""
    import importlib.util
    from medgp_amd import synth
    spec = importlib.util.spec_from_file_location("ref_fastkernel", "/root/reference/medgpc/visualization/fastkernel.py")
    fk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fk)
    for (Q, D, R, N, seed) in [(5, 2, 2, 60, 11), (5, 24, 8, 96, 12), (3, 7, 4, 70, 13)]:
        meta, t, y = synth.patient(seed, 0, D, N, interleave=(D == 7))
        th = synth.theta(seed, 0, 7, Q, D, R)
        B = np.stack(fk.compute_B_matrix(Q, D, R, th))
        mu = np.exp(th[D + Q * D * R: D + Q * D * R + Q])
        v = np.exp(th[D + Q * D * R + Q: D + Q * D * R + 2 * Q])
        dist = np.abs(t.astype(np.float64)[:, None] - t.astype(np.float64)[None, :]).reshape(-1, 1)
        K = np.zeros((N, N))
        for q in range(Q):
            kq = fk.compute_sm_1d(mu[q], v[q] ** 2, dist)[:, 0].reshape(N, N)
            K += B[q][meta[:, None], meta[None, :]] * kq
        np.savez_compressed(os.path.join(HERE, f"fastkernel_gram_Q{Q}_D{D}_R{R}.npz"), Q=Q, D=D, R=R, N=N, meta=meta, t=t, y=y,
                            theta=th, K=K)
        print(f"fastkernel gram Q={Q} D={D} R={R} N={N}: cond {np.linalg.cond(K + np.diag(np.exp(2 * th[meta]))):.2e}")


def fastkernel_univariate():
    """The single-output families (kernel_index 0 = SE, 8 = SM) composed exactly as the reference's Python does:
    SE  k(r) = compute_se_1d(s2 = exp(hyp[2]), lc = exp(hyp[1]), r)                       (vizkernel.py:317-320, fastkernel.py:50-54)
    SM  k(r) = sum_q exp(hyp[1+q]) compute_sm_1d(exp(hyp[1+Q+q]), exp(2 hyp[1+2Q+q]), r)    (vizkernel.py:347-354, fastkernel.py:33-48)
    evaluated at the pairwise distances of a time vector -> full Gram matrices without the noise diagonal."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_fastkernel", "/root/reference/medgpc/visualization/fastkernel.py")
    fk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fk)
    rng = np.random.default_rng(20241002)
    N = 50
    t = np.sort(rng.uniform(0, 200, N)).astype(np.float32)
    dist = np.abs(t.astype(np.float64)[:, None] - t.astype(np.float64)[None, :]).reshape(-1, 1)
    hyp_se = np.array([np.log(0.3), np.log(25.0), np.log(1.3)])
    K_se = fk.compute_se_1d(np.exp(hyp_se[2]), np.exp(hyp_se[1]), dist)[:, 0].reshape(N, N)
    Q = 3
    hyp_sm = np.concatenate([[np.log(0.25)], np.log(rng.uniform(0.2, 1.0, Q)), np.log(1.0 / rng.uniform(12, 72, Q)),
                             np.log(1.0 / (2 * np.pi * rng.uniform(6, 72, Q)))])
    K_sm = np.zeros((N, N))
    for q in range(Q):
        K_sm += np.exp(hyp_sm[1 + q]) * fk.compute_sm_1d(np.exp(hyp_sm[1 + Q + q]), np.exp(2 * hyp_sm[1 + 2 * Q + q]), dist)[:, 0].reshape(N, N)
    np.savez_compressed(os.path.join(HERE, "fastkernel_univariate.npz"), t=t, hyp_se=hyp_se, K_se=K_se, Q=Q, hyp_sm=hyp_sm, K_sm=K_sm)
    print("fastkernel univariate: SE", K_se.shape, "SM", K_sm.shape)


def ref_config_files():
    """The reference's OWN writers (medgpc/util/config.py:5-66, medgpc/util/binaryIO.py:6-10; both importable: os / json /
    numpy / array only) write the experiment files the C++ host parses: exp_setup.json + hyp_bound.txt for BASELINE config 1
    (scripts/feature_PT_INR.json, D = 2 -> H = 42) and for the 24-output configs (scripts/feature_all.json, H = 1114), with
    scripts/opt_prior2.json and the command line of scripts/gen_medgpc_example.sh:11 (LMC-SM, hier-gamma, Q=5, R=8 / 2,
    eta = beta_lam = 0.01); and a <alg>_mode_param.bin via write_double_to_bin.  The dictionary of path keys is the one
    run_exp_generator.py:138-163 builds (that module itself needs the cohort files, so it is not run).  Paths are relative to
    the repository root: the CPU tests run host_logic_test from there."""
    import importlib.util
    import json

    def imp(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return m
    cfg = imp("ref_config", "/root/reference/medgpc/util/config.py")
    bio = imp("ref_binaryIO", "/root/reference/medgpc/util/binaryIO.py")
    opt = json.load(open("/root/reference/scripts/opt_prior2.json"))
    for tag, feat_json, R in (("PT_INR", "feature_PT_INR.json", 2), ("all24", "feature_all.json", 8)):
        feats = [f["index"] for f in json.load(open("/root/reference/scripts/" + feat_json))["feature_list"]]
        top = f"tests/golden/ref_cfg/{tag}"
        out = os.path.join(ROOT, top)
        os.makedirs(out, exist_ok=True)
        paths = {"exp_top_dir": top, "exp_cfg_dir": top, "exp_log_dir": top + "/log", "exp_train_dir": top + "/train",
                 "exp_test_dir": top + "/test", "exp_script_dir": top + "/script", "exp_kernel_dir": top + "/kernel",
                 "exp_figure_dir": top + "/figure", "data_dir": top + "/data", "cohort_id_list": "cohort.txt",
                 "hyp_bound_file": "hyp_bound.txt"}
        cfg.write_medgpc_bound(output_dir=out, file_name="hyp_bound.txt", feature_num=len(feats), kernel_index=7,
                               mixture_num=5, rank_num=R, opt_config=opt)
        cfg.write_medgpc_config_json(exp_config_file=os.path.join(out, "exp_setup.json"), exp_path_config=paths,
                                     kernel="LMC-SM", kernel_index=7, feature_list=feats, prior="hier-gamma", prior_index=2,
                                     eta=0.01, beta_lam=0.01, mixture_num=5, rank_num=R, opt_config=opt, cv_fold_num=10,
                                     cv_assign_file=top + "/cv_assign.txt")
        print(f"ref_cfg/{tag}: D={len(feats)} H={len(feats) + 5 * (len(feats) * R + 2 + len(feats))}")
    # test-time mode kernel of a D=2, Q=3, R=2 experiment: 3 components kept -> H = 2 + 3 (2*2 + 2 + 2) = 26 doubles
    rng = np.random.default_rng(20241003)
    mode = rng.normal(0, 0.5, size=26)
    kd = os.path.join(ROOT, "tests/golden/ref_cfg/PT_INR/kernel/fold0")
    os.makedirs(kd, exist_ok=True)
    bio.write_double_to_bin(os.path.join(kd, "gmm_mode_param.bin"), mode)                  # binaryIO.py:6-10
    np.savetxt(os.path.join(kd, "gmm_mode_mixture_num.txt"), [3], fmt="%d")                # as mode_estimate.py:425
    np.save(os.path.join(ROOT, "tests/golden/ref_cfg/PT_INR/mode_expected.npy"), mode)
    print("ref_cfg/PT_INR/kernel/fold0: gmm_mode_param.bin, gmm_mode_mixture_num.txt")


def ref_feature_roundtrip():
    """The INPUT side of the file surface: per-patient feature<idx>.txt files as medgp_amd/synth_experiment.py writes them (count, then
    t / v pairs, README.md:64-72), parsed here by the reference's own reader binaryIO.load_ts_data (medgpc/util/binaryIO.py:38-43).
    The files and what the reference read from them are committed (tests/golden/ref_cfg/feature_rt); tests/test_host_files.py holds
    the C++ loader (ref: dataio/c_experiment.cpp:254-309) to them.  Three patients: 13, 40 and 2 observations over two features (the
    last one leaves feature 19 with a single observation)."""
    import importlib.util
    import shutil
    spec = importlib.util.spec_from_file_location("ref_binaryIO", "/root/reference/medgpc/util/binaryIO.py")
    bio = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bio)
    from medgp_amd.synth_experiment import make_experiment
    top = "tests/golden/ref_cfg/feature_rt"
    cwd = os.getcwd()
    os.chdir(ROOT)                       # relative paths in exp_setup.json: the CPU tests run host_logic_test from the repository root
    try:
        shutil.rmtree(top, ignore_errors=True)
        pans = ["R001", "R002", "R003"]
        ex = make_experiment(top, pans, D=2, Q=3, R=2, N=[13, 40, 2], feature_index=(18, 19), seed=11)
        out = {}
        for pan in pans:
            for fi in ex["feature_index"]:
                t, v = bio.load_ts_data(os.path.join(ex["dirs"]["data"], pan, f"feature{fi}.txt"))
                out[f"{pan}_{fi}_t"] = np.atleast_1d(t)
                out[f"{pan}_{fi}_v"] = np.atleast_1d(v)
        out["stats"] = np.array(ex["stats"])
        np.savez(os.path.join(top, "parsed_by_reference.npz"), **out)
        for d in ("train", "test", "kernel"):
            shutil.rmtree(os.path.join(top, d), ignore_errors=True)     # empty directories: nothing to commit
        print("ref_cfg/feature_rt:", {k: v.shape for k, v in out.items()})
    finally:
        os.chdir(cwd)


def ref_prior():
    """tests/golden/ref_prior.json.gz, ref_prior_inference.json.gz: outputs of the reference's OWN compiled code -- the three translation
    units of /root/reference/medgpc/src that build unmodified with plain g++ (prior/c_prior.cpp, core/c_hyperparam.cpp,
    inference/c_inference_prior.cpp; no MKL, no rapidjson, no stand-in headers), built by `make -C oracle ref` into oracle/_ref/ and
    driven by oracle/ref_prior_dump.cpp and oracle/ref_prior_inference_dump.cpp (whose headers say exactly what is and is not the
    reference in each).  Pins rows a2 (theta split), a19 (prior terms) and the variational-EM start state of f1."""
    import json
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], stdout=subprocess.DEVNULL)
    for exe, name in (("ref_prior_dump", "ref_prior.json"), ("ref_prior_inference_dump", "ref_prior_inference.json")):
        txt = subprocess.run([os.path.join(ROOT, "oracle", "_ref", exe)], check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout
        d = json.loads(txt)            # (must parse)
        import gzip
        with gzip.GzipFile(os.path.join(HERE, name + ".gz"), "wb", mtime=0) as f:    # (mtime 0: the same bytes on every run)
            f.write(txt.encode())
        print(name + ".gz", len(txt), "bytes of JSON;", list(d.keys()))


if __name__ == "__main__":
    which = sys.argv[1:] or ["appendix_a", "fastkernel", "fastkernel_gram", "fastkernel_univariate", "config5", "ref_config_files", "ref_feature_roundtrip", "ref_prior"]
    for w in which:
        globals()[w]()
