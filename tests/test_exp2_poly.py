"""CPU check of the polynomial inside exp2_nonpos (medgp_amd/csrc/kernels_assemble.h): the coefficients are read from the header
and the polynomial is compared with 2^f in 40-digit arithmetic on [-1/2, 1/2].  Guards the pair kernels' only transcendental
against an edited digit (their GPU parity tests would still pass at 1e-11 with a polynomial that is 1000 times worse)."""
import os
import re

import mpmath as mp

HDR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "medgp_amd", "csrc", "kernels_assemble.h")


def _coefficients():
    src = open(HDR).read()
    body = src[src.index("double exp2_nonpos(double z)"):]
    body = body[:body.index("return ldexp")]
    lead = re.search(r"double p = ([0-9.eE+-]+);", body).group(1)
    rest = re.findall(r"p = fma\(p, f, ([0-9.eE+-]+)\);", body)
    return [float(lead)] + [float(c) for c in rest]          # descending powers


def test_exp2_polynomial_accuracy_and_exact_one():
    c = _coefficients()
    assert len(c) == 11 and c[-1] == 1.0                      # degree 10, p(0) = 1 exactly (K's diagonal: dt = 0)
    mp.mp.dps = 40
    worst = mp.mpf(0)
    for i in range(4001):
        f = mp.mpf(-0.5) + mp.mpf(i) / 4000
        p = mp.mpf(0)
        for ck in c:
            p = p * f + mp.mpf(ck)
        worst = max(worst, abs(p - mp.power(2, f)) / mp.power(2, f))
    assert worst < 5e-16, float(worst)                        # measured 3.0e-16 (scratch/exp2_minimax.py)


def test_exp2_polynomial_in_double_arithmetic():
    # the same Horner chain in IEEE doubles (no fma here: one more rounding per step than the device) stays within 4 ulp
    c = _coefficients()
    worst = 0.0
    for i in range(2001):
        f = -0.5 + i / 2000.0
        p = 0.0
        for ck in c:
            p = p * f + ck
        ref = float(mp.power(2, mp.mpf(f)))
        worst = max(worst, abs(p - ref) / ref)
    assert worst < 1e-15, worst
