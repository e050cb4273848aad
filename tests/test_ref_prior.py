"""Rows a2 / a19 / f1 pinned to the REFERENCE's OWN COMPILED CODE (round 6).

Three translation units of /root/reference/medgpc/src compile unmodified with plain g++ (no MKL, no rapidjson, no stand-in
headers): prior/c_prior.cpp, core/c_hyperparam.cpp, inference/c_inference_prior.cpp.  `make -C oracle ref` builds them in the build
container, oracle/ref_prior_dump.cpp and oracle/ref_prior_inference_dump.cpp drive them (the second one supplies the dense
evaluation the prior stage starts from -- its header says exactly what that does and does not pin), and
tests/golden/make_golden.py::ref_prior commits what they print as tests/golden/ref_prior.json.gz / ref_prior_inference.json.gz.

Held to those numbers here, exactly (integers, float parameters) or to 1e-15 relative (lp, dlp: one or two roundings of libm's log):
  * oracle.Prior.hier_gamma, synth.hier_gamma_prior, optimizer_oracle.VarEMPrior      (what every parity test builds its priors with)
  * the C++ host's c_prior (medgp_host.cpp; what medgp_train / medgp_test upload)      via `host_logic_test priordump`
  * the oracle's prior log-densities and prior stage (medgp_oracle.c apply_prior)      -> the device's k_epilogue is held to the same
    reference numbers in tests/test_parity_gpu.py::test_device_prior_stage_vs_reference_compiled_prior
  * the theta split [lik | cov | mean]                                                 (c_hyperparam::set_hyp_all / get_hyp_all)
"""
import ctypes as C
import gzip
import json
import os
import subprocess

import numpy as np
import pytest

from medgp_amd import synth
from oracle import optimizer_oracle as OO
from oracle import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _load(name):
    with gzip.open(os.path.join(HERE, "golden", name), "rt") as f:
        return json.load(f)


@pytest.fixture(scope="module")
def ref():
    return _load("ref_prior.json.gz")


@pytest.fixture(scope="module")
def ref_inf():
    return _load("ref_prior_inference.json.gz")


def _theta_order(p, H, nlik):
    """(flag, type, exp, p0, p1) in theta order [lik | cov] from one dumped c_prior; absent parameters -> (0, 1) as the flat arrays hold them"""
    flag = np.array(p["flag_lik"] + p["flag_cov"], np.uint8)
    typ = np.array(p["type_lik"] + p["type_cov"], np.int32)
    ex = np.array(p["exp_lik"] + p["exp_cov"], np.uint8)
    ln = np.array(p["fix_lik_len"] + p["fix_cov_len"])
    p0 = np.where(ln > 0, np.array(p["fix_lik_p0"] + p["fix_cov_p0"]), 0.0).astype(np.float32)
    p1 = np.where(ln > 1, np.array(p["fix_lik_p1"] + p["fix_cov_p1"]), 1.0).astype(np.float32)
    assert flag.shape[0] == H and len(p["flag_lik"]) == nlik and not p["flag_mean"]
    return flag, typ, ex, p0, p1


def test_fixture_is_what_the_reference_sources_say(ref):
    """Sanity of the fixture itself against the text of prior/c_prior.cpp:222-279 (so that a broken driver cannot pin nonsense)."""
    assert ref["eta"] == pytest.approx(0.01, rel=1e-7) and ref["beta_lam"] == pytest.approx(0.01, rel=1e-7)
    assert [(s["Q"], s["D"], s["R"]) for s in ref["shapes"]] == [(5, 2, 2), (5, 24, 8), (5, 64, 8)]
    for s in ref["shapes"]:
        Q, D, R = s["Q"], s["D"], s["R"]
        m2 = s["mode2"]
        assert s["ncov"] == Q * (D * R + 2 + D) and s["nlik"] == D and s["nmean"] == 0
        assert m2["type_cov"][:Q * D * R] == [1] * (Q * D * R) and m2["type_cov"][-Q * D:] == [2] * (Q * D)
        assert m2["cov_varEM"] == [1.0] * (2 * Q * (D * R + R))
        assert m2["cov_varEM_fix"][:4] == [0.5] * 4 and m2["cov_varEM_fix"][4] == float(np.float32(0.01))
        assert s["mode2_default"]["cov_varEM_fix"][4] == 50.0 and s["mode2_default"]["fix_cov_p1"][-1] == 0.5
        assert not any(s["mode0"]["flag_cov"]) and s["mode0"]["cov_varEM"] == []
        assert not any(s["mode2_kernel0"]["flag_cov"])        # kernel index != 7: "prior will not be effective"


def test_python_prior_builders_match_the_compiled_reference(ref):
    for s in ref["shapes"]:
        Q, D, R = s["Q"], s["D"], s["R"]
        H = D + s["ncov"]
        flag, typ, ex, p0, p1 = _theta_order(s["mode2"], H, D)
        # oracle.Prior.hier_gamma: flag / type / exp exactly; parameters wherever a prior is active
        pr = O.Prior.hier_gamma(Q, D, R, 0.01, 0.01)
        assert np.array_equal(pr.flag, flag) and np.array_equal(pr.exp, ex)
        assert np.array_equal(pr.type[flag == 1], typ[flag == 1])
        assert np.array_equal(pr.p0[flag == 1], p0[flag == 1]) and np.array_equal(pr.p1[flag == 1], p1[flag == 1])
        # inactive hypers: the reference leaves type -1 and no parameters
        assert (typ[flag == 0] == -1).all() and (pr.type[flag == 0] == -1).all()
        # synth.hier_gamma_prior (bench.py, every GPU test)
        f2, t2, e2, a2, b2 = synth.hier_gamma_prior(Q, D, R, 0.01)
        assert np.array_equal(f2, flag) and np.array_equal(e2, ex) and np.array_equal(t2, typ)
        assert np.array_equal(a2[flag == 1], p0[flag == 1]) and np.array_equal(b2[flag == 1], p1[flag == 1])
        # the variational-EM start state (row f1)
        vp = OO.VarEMPrior(Q, D, R, 0.01)
        assert vp.cov_varEM == s["mode2"]["cov_varEM"]
        assert [float(v) for v in vp.fix] == s["mode2"]["cov_varEM_fix"]          # float(np.float32(0.01)) in slot 4
        assert vp.type_A == s["mode2"]["type_cov"][:Q * D * R]
        assert [float(v) for v in vp.var_A] == s["mode2"]["fix_cov_p1"][:Q * D * R]


def test_host_c_prior_matches_the_compiled_reference(ref):
    exe = os.path.join(ROOT, "medgp_amd", "host", "host_logic_test")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.dirname(exe)], stdout=subprocess.DEVNULL)
    out = subprocess.run([exe, "priordump"], check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout
    host = json.loads(out)
    keys = ["flag_lik", "flag_cov", "flag_mean", "exp_lik", "exp_cov", "exp_mean", "type_lik", "type_cov", "type_mean",
            "fix_lik_len", "fix_lik_p0", "fix_lik_p1", "fix_cov_len", "fix_cov_p0", "fix_cov_p1", "fix_mean_len", "cov_varEM", "cov_varEM_fix"]
    for hs, rs in zip(host["shapes"], ref["shapes"]):
        assert (hs["Q"], hs["D"], hs["R"]) == (rs["Q"], rs["D"], rs["R"])
        for case in ("mode0", "mode2", "mode2_default", "mode2_kernel0", "mode2_test", "mode0_test"):
            for k in keys:
                assert hs[case][k] == rs[case][k], (rs["Q"], rs["D"], case, k)
        assert hs["test_flag_theta_order"] == rs["test_flag_theta_order"] and hs["test_type_theta_order"] == rs["test_type_theta_order"]
        # the clamp hit exactly the A entries whose mode value is 0.0, and nothing else changed
        D, nA = rs["D"], rs["Q"] * rs["D"] * rs["R"]
        zero = [i - D for i in range(D, D + nA) if rs["test_mode"][i] == 0.0]
        assert zero and [i for i, t in enumerate(rs["mode2_test"]["type_cov"]) if t == 0] == zero
        # what medgp_set_prior receives (flatten) is the reference's state in theta order
        H = D + rs["ncov"]
        flag, typ, ex, p0, p1 = _theta_order(rs["mode2_test"], H, D)
        assert hs["flat_flag"] == flag.tolist() and hs["flat_type"] == typ.tolist() and hs["flat_exp"] == ex.tolist()
        act = (flag == 1) & (typ > 0)
        assert np.array_equal(np.array(hs["flat_p0"], np.float32)[act], p0[act]) and np.array_equal(np.array(hs["flat_p1"], np.float32)[act], p1[act])


def test_oracle_prior_densities_match_the_compiled_reference(ref):
    lib = O.lib()
    lib.medgp_oracle_prior_lik.argtypes = [C.c_int, C.c_double, C.c_float, C.c_float, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.medgp_oracle_prior_lik.restype = None
    g = ref["lik_grid"]
    xs = g["x"]
    n = len(g["p0"])
    assert n == 5 * len(xs) and 0.0 in xs and 0.25 in xs and -1.0 in xs       # the grid hits x == m (the Laplace kink) for every one of the five parameter pairs
    worst = 0.0
    for i in range(n):
        x, p0, p1 = xs[i % len(xs)], g["p0"][i], g["p1"][i]
        for typ, klp, kd in ((1, "normal_lp", "normal_dlp"), (2, "laplace_lp", "laplace_dlp")):
            lp, dlp = C.c_double(), C.c_double()
            lib.medgp_oracle_prior_lik(typ, x, p0, p1, O.REF_PI, C.byref(lp), C.byref(dlp))
            for got, want in ((lp.value, g[klp][i]), (dlp.value, g[kd][i])):
                err = abs(got - want) / max(abs(want), 1e-300) if want != 0.0 else abs(got)
                worst = max(worst, err)
                assert err <= 1e-15, (typ, x, p0, p1, got, want)
    # the kink: dlp == 0 exactly at x == location, -/+ 1/scale either side (c_prior.cpp:404-417)
    k = [i for i in range(n) if xs[i % len(xs)] == g["p0"][i]]
    assert len(k) == 5 and all(g["laplace_dlp"][i] == 0.0 for i in k)
    assert all(g["laplace_dlp"][i - 1] == 1.0 / g["p1"][i] and g["laplace_dlp"][i + 1] == -1.0 / g["p1"][i] for i in k)
    # get_one_lik_cov through a whole object: indices of an A entry (normal), a clamped A entry, mu (no prior), kappa (laplace)
    for s in ref["shapes"]:
        Q, D, R = s["Q"], s["D"], s["R"]
        flag, typ, ex, p0, p1 = _theta_order(s["mode2_test"], D + s["ncov"], D)
        it = iter(zip(s["lik_cov_lp"], s["lik_cov_dlp"]))
        for idx in s["lik_cov_idx"]:
            for x in s["lik_cov_x"]:
                wlp, wdlp = next(it)
                h = D + idx
                lp, dlp = C.c_double(), C.c_double()
                lib.medgp_oracle_prior_lik(int(typ[h]), x, float(p0[h]), float(p1[h]), O.REF_PI, C.byref(lp), C.byref(dlp))
                assert abs(lp.value - wlp) <= 1e-15 * max(1.0, abs(wlp)) and abs(dlp.value - wdlp) <= 1e-15 * max(1.0, abs(wdlp)), (idx, x)


def test_theta_split_matches_c_hyperparam(ref):
    """c_hyperparam::set_hyp_all: [lik | cov | mean] (ref: core/c_hyperparam.cpp:99-122) -- the layout include/medgp_hip.h documents
    and medgp_oracle_num_lik / num_cov implement."""
    for s in ref["shapes"]:
        Q, D, R = s["Q"], s["D"], s["R"]
        nl, nc = O.num_lik(7, D), O.lib().medgp_oracle_num_cov(7, Q, D, R)
        th = s["hyp_in"]
        assert (nl, nc) == (s["nlik"], s["ncov"]) and s["hyp_counts"] == [nl, nc, 2, nl + nc + 2]
        assert s["hyp_lik"] == th[:nl] and s["hyp_cov"] == th[nl:nl + nc] and s["hyp_mean"] == th[nl + nc:]
        assert s["hyp_all"] == th                                         # get_hyp_all re-assembles the same order
        assert O.num_hyp(7, Q, D, R) == nl + nc and synth.num_hyp(7, Q, D, R) == nl + nc


def test_oracle_prior_stage_matches_the_compiled_c_inference_prior(ref_inf):
    """inference/c_inference_prior.cpp:60-150 (compiled from the reference; the dense evaluation it starts from is handed in by
    the driver): nlml - sum lp, dnlml - hyp * dlp (exp chain rule) or - dlp, clamp -> 0, untouched on failure / without gradient."""
    lib = O.lib()
    dp, u8p, i32p, fp = C.POINTER(C.c_double), C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.POINTER(C.c_float)
    lib.medgp_oracle_apply_prior.argtypes = [C.c_int, dp, C.c_double, C.c_int, u8p, i32p, u8p, fp, fp, dp, dp]
    lib.medgp_oracle_apply_prior.restype = None
    names = [c["name"] for c in ref_inf["cases"]]
    assert len(names) == 10 and "PT_INR_mode2_failed" in names and "D64_mode2_varem" in names
    for c in ref_inf["cases"]:
        H = len(c["theta"])
        if not c["base_ok"]:                                              # c_inference_exact failed: nothing is touched (:58)
            assert c["ok"] == 0 and c["nlml"] == -777.0 and set(c["dnlml"]) == {-777.0}
            continue
        assert c["ok"] == 1
        hval = np.array(c["hval"])
        # the transformed values the driver formed are exp(theta) behind the A block, theta itself inside it
        nl, nA = c["D"], c["Q"] * c["D"] * c["R"]
        th = np.array(c["theta"])
        want_h = np.exp(th)
        want_h[nl:nl + nA] = th[nl:nl + nA]
        np.testing.assert_allclose(hval, want_h, rtol=5e-16, atol=0)     # (numpy's exp and the C library's may differ in the last bit)
        assert np.array_equal(hval[nl:nl + nA], th[nl:nl + nA])
        flag = np.array(c["prior_flag"], np.uint8)
        typ = np.array(c["prior_type"], np.int32)
        ex = np.array(c["prior_exp"], np.uint8)
        p0 = np.array(c["prior_p0"], np.float32)
        p1 = np.array(c["prior_p1"], np.float32)
        nlml = C.c_double(c["base_nlml"])
        grad = np.array(c["base_dnlml"])
        lib.medgp_oracle_apply_prior(H, hval.ctypes.data_as(dp), O.REF_PI, int(c["flag_grad"]), flag.ctypes.data_as(u8p), typ.ctypes.data_as(i32p),
                                     ex.ctypes.data_as(u8p), p0.ctypes.data_as(fp), p1.ctypes.data_as(fp), C.byref(nlml), grad.ctypes.data_as(dp))
        assert abs(nlml.value - c["nlml"]) <= 4e-16 * H * max(1.0, abs(c["nlml"])), (c["name"], nlml.value, c["nlml"])   # same terms, same order: a few ulp at most
        if c["flag_grad"]:
            want = np.array(c["dnlml"])
            assert np.all(np.abs(grad - want) <= 1e-15 * np.maximum(1.0, np.abs(want))), c["name"]
            clamp = (flag == 1) & (typ == 0)
            assert (want[clamp] == 0.0).all() and (grad[clamp] == 0.0).all()
            if "varem" in c["name"] or "testclamp" in c["name"]:
                assert clamp.any(), c["name"]
        else:
            assert set(c["dnlml"]) == {-777.0}                            # no gradient asked: dnlml untouched


@pytest.mark.skipif(not os.path.isdir("/root/reference/medgpc/src"), reason="the reference tree exists in the build container only")
def test_fixtures_regenerate_from_the_reference_sources():
    """Where /root/reference exists: rebuild oracle/_ref from the reference's sources as they lie (make -C oracle ref: plain g++, no
    stand-in headers) and run both drivers -- the committed fixtures are exactly what they print."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], stdout=subprocess.DEVNULL)
    for exe, name in (("ref_prior_dump", "ref_prior.json.gz"), ("ref_prior_inference_dump", "ref_prior_inference.json.gz")):
        out = subprocess.run([os.path.join(ROOT, "oracle", "_ref", exe)], check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
        with gzip.open(os.path.join(HERE, "golden", name), "rb") as f:
            assert f.read() == out, name
    # the three translation units come from the reference tree itself, compiled with nothing but its own include path
    mk = open(os.path.join(ROOT, "oracle", "Makefile")).read()
    assert "$(REF)/prior/c_prior.cpp" in mk and "$(REF)/core/c_hyperparam.cpp" in mk and "$(REF)/inference/c_inference_prior.cpp" in mk
    assert "-I$(REF)" in mk and "mkl" not in mk.split("REFCXXFLAGS")[1].split("\n")[0].lower()
