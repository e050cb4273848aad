"""near-minimax polynomials for 2^f on [-1/2, 1/2] with p(0) = 1 exactly (Remez exchange on the relative error, mpmath);
prints double coefficients and the max relative error of the ROUNDED polynomial, next to the degree-11 Taylor form in use"""
import mpmath as mp
mp.mp.dps = 60
LN2 = mp.log(2)
def relerr(coef, f):   # coef[0] = 1, ascending
    p = mp.mpf(0)
    for c in reversed(coef): p = p * f + c
    return (p - mp.power(2, f)) / mp.power(2, f)
def remez(n, iters=12):
    # p(f) = 1 + f * q(f), q of degree n-1: n unknowns + the level E -> n+1 reference points
    m = n + 1
    xs = [mp.mpf('0.5') * mp.cos(mp.pi * (2 * i + 1) / (2 * m)) for i in range(m)][::-1]
    for it in range(iters):
        A = mp.matrix(m, m); rhs = mp.matrix(m, 1)
        for r, x in enumerate(xs):
            w = mp.power(2, x)
            for c in range(n): A[r, c] = x ** (c + 1) / w
            A[r, n] = (-1) ** r
            rhs[r] = (w - 1) / w
        sol = mp.lu_solve(A, rhs)
        coef = [mp.mpf(1)] + [sol[c] for c in range(n)]
        # new reference: extrema of the error on a fine grid, one per sign-alternating interval
        grid = [mp.mpf(-0.5) + mp.mpf(i) / 4000 for i in range(4001)]
        ev = [relerr(coef, g) for g in grid]
        ext = []
        cur_sign = None; best = None
        for g, e in zip(grid, ev):
            s = 1 if e >= 0 else -1
            if s != cur_sign:
                if best is not None: ext.append(best)
                cur_sign = s; best = (abs(e), g)
            elif abs(e) > best[0]: best = (abs(e), g)
        ext.append(best)
        if len(ext) < m: break
        ext = sorted(sorted(ext, reverse=True)[:m], key=lambda t: t[1]) if len(ext) > m else ext
        xs = [g for _, g in ext]
    return coef
def maxerr(coefd):
    c = [mp.mpf(x) for x in coefd]
    return max(abs(relerr(c, mp.mpf(-0.5) + mp.mpf(i) / 20000)) for i in range(20001))
taylor = [float(LN2 ** k / mp.factorial(k)) for k in range(12)]
print("Taylor degree 11 (in use): max rel err %.3e" % float(maxerr(taylor)))
for n in (9, 10):
    coef = remez(n)
    cd = [float(c) for c in coef]
    print("minimax degree %d: max rel err of the double-rounded polynomial %.3e" % (n, float(maxerr(cd))))
    print("   coefficients (ascending):", ", ".join(repr(c) for c in cd))
