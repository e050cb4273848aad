"""GPU test of the C++ host adapter (medgp_amd/host): the object set and call sequence of
main_one_train.cpp:103-118,228-238 and main_one_test.cpp:386-399, checked against the oracle."""
import os
import struct
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from medgp_amd import synth
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "medgp_amd", "host", "host_selftest")


@pytest.mark.parametrize("D,N,Q,R,prior_mode", [(2, 150, 5, 2, 0), (6, 200, 3, 4, 2)])
def test_cpp_host_adapter_matches_oracle(tmp_path, built_lib, D, N, Q, R, prior_mode):
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-C", os.path.dirname(EXE)])
    m, t, y = synth.patient(42, 0, D, N)
    th = synth.theta(42, 0, 7, Q, D, R)
    m2 = np.array([0, D - 1, 1 % D], np.int32)
    t2 = np.array([5.0, 100.5, float(t[3])], np.float32)
    H = th.size
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        f.write(struct.pack("7i", D, N, Q, R, H, prior_mode, m2.size))
        f.write(m.tobytes()); f.write(t.tobytes()); f.write(y.tobytes()); f.write(th.tobytes())
        f.write(m2.tobytes()); f.write(t2.tobytes())
    r = subprocess.run([EXE, str(fin), str(fout)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    b = open(fout, "rb").read()
    ok = struct.unpack_from("i", b, 0)[0]
    nlml = struct.unpack_from("d", b, 4)[0]
    grad = np.frombuffer(b, np.float64, H, 12)
    o = 12 + 8 * H
    mean = np.frombuffer(b, np.float32, m2.size, o); o += 4 * m2.size
    var = np.frombuffer(b, np.float32, m2.size, o); o += 4 * m2.size
    alpha = np.frombuffer(b, np.float32, N, o); o += 4 * N
    beta = struct.unpack_from("f", b, o)[0]
    pr = O.Prior.hier_gamma(Q, D, R, 0.01, 0.01) if prior_mode == 2 else None
    ref = O.nlml_grad(7, Q, D, R, m, t, y, th, prior=pr, want_alpha=True)
    assert ok == 1
    assert abs(nlml - ref["nlml"]) <= 1e-10 * abs(ref["nlml"])
    gs = np.abs(ref["grad"]).max()
    assert np.all(np.abs(grad - ref["grad"]) <= 1e-6 * np.maximum(np.abs(ref["grad"]), 1e-3 * gs))
    rp = O.fit_predict(7, Q, D, R, m, t, y, th, m2, t2)
    np.testing.assert_allclose(mean, rp["mean"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(var, rp["var"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(alpha, ref["alpha"], rtol=1e-5, atol=1e-6 * np.abs(ref["alpha"]).max())
    assert abs(beta - ref["beta"]) <= 1e-5 * abs(ref["beta"])
