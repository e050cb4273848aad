"""The product's resumable optimiser state machines (medgp_amd/host/medgp_optimizer.cpp) against the independent, loop
structured Python restatements of the reference's optimisers (oracle/optimizer_oracle.py; ref:
util/c_optimizer_scg.cpp:25-284, util/c_optimizer_varEM.cpp:26-206) on analytic objectives.  The objectives use only
+ - * / sqrt log, so both sides evaluate them identically and the comparison is BIT FOR BIT: evaluation counts, the
returned loss, every hyper, the variational state (psi, delta, phi, tau) and the clamp decisions.  CPU only."""
import math
import os
import subprocess

import numpy as np
import pytest

from oracle import optimizer_oracle as OO

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "medgp_amd", "host")
EXE = os.path.join(HOST, "host_logic_test")


def quad(x):
    f, g = 0.0, [0.0] * len(x)
    for i in range(len(x)):
        w = 1.0 + 3.0 * i
        f += 0.5 * w * (x[i] - 0.3 * i) * (x[i] - 0.3 * i)
        g[i] = w * (x[i] - 0.3 * i)
    return True, f, g


def rosen(x):
    f, g = 0.0, [0.0] * len(x)
    for i in range(len(x) - 1):
        a, b = x[i + 1] - x[i] * x[i], 1 - x[i]
        f += 100 * a * a + b * b
        g[i] += -400 * a * x[i] - 2 * b
        g[i + 1] += 200 * a
    return True, f, g


def hole(x):
    if x[0] > 2.5:
        return False, 0.0, []
    ok, f, g = quad(x)
    if x[1] > 4.0:
        f = float("nan")
    return ok, f, g


def poly(x):
    n = len(x)
    f, g = 0.0, [0.0] * n
    total = 0.0
    for i in range(n):
        total += x[i]
    for i in range(n):
        w, c = 1.0 + 0.5 * float(i), 0.25 * float(i) - 1.0
        d = x[i] - c
        f += 0.5 * w * d * d + 0.05 * d * d * d * d
        g[i] = w * d + 0.2 * d * d * d
    f += 0.05 * total * total
    for i in range(n):
        g[i] += 0.1 * total
    return True, f, g


OBJ = {0: quad, 1: rosen, 2: hole, 3: poly}
INIT = {0: [2.0] * 6, 1: [-1.2, 1.0, -0.5, 0.8], 2: [2.4, 3.9, 0.0], 3: [1.5, -2.0, 0.7, 3.0, -0.4]}


@pytest.fixture(scope="module")
def dump(tmp_path_factory):
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-s", "-C", HOST, "host_logic_test"])
    out = tmp_path_factory.mktemp("opt") / "opt.bin"
    r = subprocess.run([EXE, "optdump", str(out)], capture_output=True, text=True)
    assert r.returncode == 0 and "OPTDUMP ok" in r.stdout, r.stdout + r.stderr
    return np.fromfile(out, np.float64)


def test_scg_machine_equals_reference_restatement(dump):
    pos = 0
    for _ in range(7):
        oid, budget, nev, loss = int(dump[pos]), int(dump[pos + 1]), int(dump[pos + 2]), dump[pos + 3]
        n = len(INIT[oid])
        X = dump[pos + 4: pos + 4 + n]
        pos += 4 + n
        l2, X2, n2 = OO.scg(budget, INIT[oid], OBJ[oid])
        assert n2 == nev, (oid, budget, n2, nev)
        assert l2 == loss, (oid, budget, l2, loss)
        assert np.array_equal(np.array(X2), X), (oid, budget)
        if budget < 0:
            assert nev <= -budget        # a negative budget counts function evaluations (ref :73,88,114,234)


def test_varem_machine_equals_reference_restatement(dump):
    pos = 0
    for _ in range(7):
        pos += 4 + len(INIT[int(dump[pos])])
    Q, D, R = 2, 2, 2
    nlik, H = D, D + Q * (D * R + 2 + D)
    PI = 3.14159265
    for variant in range(2):
        assert int(dump[pos]) == 100 + variant
        nev, loss = int(dump[pos + 1]), dump[pos + 2]
        X = dump[pos + 3: pos + 3 + H]
        ncv = 2 * Q * (D * R + R)
        cv = dump[pos + 3 + H: pos + 3 + H + ncv]
        types = dump[pos + 3 + H + ncv: pos + 3 + H + ncv + Q * D * R]
        pos += 3 + H + ncv + Q * D * R
        init = [0.3 * float((h * 7) % 5) - 0.6 for h in range(H)]
        if variant:
            init[nlik + 1] = 0.0
            init[nlik + 6] = 0.0
        prior = OO.VarEMPrior(Q, D, R, 0.3 if variant else 0.01)
        count = [0]

        def obj_of_prior(pr, variant=variant):
            def obj(th):
                ok, f, g = poly(th)
                if variant:
                    g[nlik + 1] = 0.0
                    g[nlik + 6] = 0.0
                for a in range(Q * D * R):
                    h = nlik + a
                    if pr.type_A[a] == 0:
                        g[h] = 0.0
                        continue
                    mean, var = 0.0, float(pr.var_A[a])
                    lp = -1.0 * (th[h] - mean) * (th[h] - mean) / (2.0 * var) - math.log(2 * PI * var) / 2.0
                    f -= lp
                    g[h] -= -1.0 * (th[h] - mean) / var
                count[0] += 1
                return True, f, g
            return obj

        l2, X2, trace = OO.varem(-6, init, obj_of_prior, prior, nlik, 15)
        assert count[0] == nev, (variant, count[0], nev)
        assert l2 == loss, (variant, l2, loss)
        assert np.array_equal(np.array(X2), X), variant
        assert np.array_equal(np.array(prior.cov_varEM), cv), variant
        assert np.array_equal(np.array(prior.type_A, dtype=np.float64), types), variant
        if variant:
            assert types[1] == 0 and types[6] == 0 and X[nlik + 1] == 0.0 and X[nlik + 6] == 0.0   # psi == 0 clamps (ref :151-154)
