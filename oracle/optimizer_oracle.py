"""Plain restatements of the reference's optimisers (TEST INFRASTRUCTURE ONLY, like everything under oracle/).

    scg(...)    <- c_optimizer_scg::optimize    ref: medgpc/src/util/c_optimizer_scg.cpp:25-284
    varem(...)  <- c_optimizer_varEM::optimize  ref: medgpc/src/util/c_optimizer_varEM.cpp:26-206

Non-resumable, loop structured exactly like the reference (the product's scg_machine / varem_machine in
medgp_amd/host/medgp_optimizer.cpp are resumable state machines of the same algorithms; tests compare the two).
`obj(theta) -> (ok, f, grad)` plays c_objective::compute_objective(true, ...): on ok == False the caller's f3 / df3 keep
their previous values (ref: util/c_objective_one.cpp:40-82 returns before writing its outputs).
Dot products are plain ascending fp64 loops (the reference calls MKL cblas_ddot, whose summation order is unspecified;
the product uses the same ascending loop, so the two agree bit for bit on identical objective values).
"""
import math

import numpy as np


def _dot(a, b):
    s = 0.0
    for x, y in zip(a, b):
        s += x * y
    return s


def _signbit(v):
    return 1 if math.copysign(1.0, float(v)) < 0 else 0


def scg(max_iteration, init_parameter, obj):
    """ref: c_optimizer_scg.cpp:25-284.  Returns (opt_loss, opt_parameter, evaluations).
    A negative max_iteration is an EVALUATION budget: i advances by signbit(max_iteration) per objective call
    (ref :73, :88, :114, :234).  A positive one would never advance i in the reference (its outer loop adds
    signbit() = 0, :88); it is treated as a line-search count like minimize.m, as the product does."""
    INT, EXT, MAX, RATIO, SIG = 0.1, 3.0, 20, 10.0, 0.1          # ref :37-42
    RHO = SIG / 2.0
    red = 1.0
    sb = _signbit(max_iteration)
    n_eval = 0
    i = 0
    x1 = x2 = x4 = d1 = d2 = d4 = f1 = f2 = f4 = 0.0
    init = [float(v) for v in init_parameter]
    ok, f0, df0 = obj(init)                                        # ref :67-71
    n_eval += 1
    df0 = [float(v) for v in df0]
    i += sb                                                        # ref :75
    s = [-1.0 * g for g in df0]                                    # ref :77-80
    d0 = -1.0 * _dot(s, s)                                         # ref :82-83
    x3 = red / (1.0 - d0)                                          # ref :84
    opt_loss, X = f0, list(init)                                   # ref :87-88
    f3, df3, d3 = f0, list(df0), 0.0
    obj_flag = ok
    while i < abs(max_iteration):                                  # ref :89
        i += sb + (1 if max_iteration > 0 else 0)                  # ref :90 (+ the minimize.m count for positive budgets)
        X0, F0, dF0 = list(X), opt_loss, list(df0)                 # ref :92-94
        M = MAX if max_iteration > 0 else min(int(MAX), abs(max_iteration) - i)   # ref :96-101
        while True:                                                # ref :103
            x2, f2, d2 = 0.0, opt_loss, d0                         # ref :104-107
            f3, df3 = opt_loss, list(df0)                          # ref :109-110
            success = False
            while (not success) and M > 0:                         # ref :114
                M -= 1
                i += sb
                newp = [X[j] + x3 * s[j] for j in range(len(X))]   # ref :118-121
                ok, ft, gt = obj(newp)
                n_eval += 1
                obj_flag = ok
                if ok:
                    f3, df3 = ft, [float(v) for v in gt]
                if (not ok) or math.isinf(f3) or math.isnan(f3):   # ref :127-132
                    x3 = (x2 + x3) / 2.0
                else:
                    success = True
            if f3 < F0:                                            # ref :135-141
                X0 = [X[j] + x3 * s[j] for j in range(len(X))]
                F0, dF0 = f3, list(df3)
            d3 = _dot(df3, s)                                      # ref :143-149
            if d3 > SIG * d0 or f3 > opt_loss + x3 * RHO * d0 or M == 0:   # ref :152-154
                break
            x1, f1, d1 = x2, f2, d2                                # ref :157-163
            x2, f2, d2 = x3, f3, d3
            A = 6.0 * (f1 - f2) + 3.0 * (d2 + d1) * (x2 - x1)      # ref :165-166
            B = 3.0 * (f2 - f1) - (2.0 * d1 + d2) * (x2 - x1)
            temp = B * B - A * d1 * (x2 - x1)
            if temp < 0:                                           # ref :168-182
                x3 = x2 * EXT
            else:
                den = B + math.sqrt(temp)
                x3 = x1 - (d1 * (x2 - x1) ** 2.0 / den) if den != 0.0 else float("nan")
                if math.isnan(x3) or math.isinf(x3) or x3 < 0:
                    x3 = x2 * EXT
                elif x3 > x2 * EXT:
                    x3 = x2 * EXT
                elif x3 < x2 + INT * (x2 - x1):
                    x3 = x2 + INT * (x2 - x1)
        while (abs(d3) > -1.0 * SIG * d0 or f3 > opt_loss + x3 * RHO * d0) and M > 0:   # ref :185
            if d3 > 0 or f3 > opt_loss + x3 * RHO * d0:            # ref :186-195
                x4, f4, d4 = x3, f3, d3
            else:
                x2, f2, d2 = x3, f3, d3
            if f4 > opt_loss:                                      # ref :197-202
                den = f4 - f2 - d2 * (x4 - x2)
                x3 = x2 - (0.5 * d2 * (x4 - x2) ** 2.0) / den if den != 0.0 else float("nan")
                if math.isnan(x3) or math.isinf(x3):
                    x3 = (x2 + x4) / 2.0
            else:                                                  # ref :203-216
                A = 6.0 * (f2 - f4) / (x4 - x2) + 3.0 * (d4 + d2)
                B = 3.0 * (f4 - f2) - (2.0 * d2 + d4) * (x4 - x2)
                disc = B * B - A * d2 * (x4 - x2) ** 2.0
                if disc < 0:
                    x3 = (x2 + x4) / 2.0
                else:
                    x3 = x2 + (math.sqrt(disc) - B) / A if A != 0.0 else float("nan")
                    if math.isnan(x3) or math.isinf(x3):
                        x3 = (x2 + x4) / 2.0
            x3 = max(min(x3, x4 - INT * (x4 - x2)), x2 + INT * (x4 - x2))   # ref :217
            newp = [X[j] + x3 * s[j] for j in range(len(X))]
            ok, ft, gt = obj(newp)                                 # ref :223-226
            n_eval += 1
            obj_flag = ok
            if ok:
                f3, df3 = ft, [float(v) for v in gt]
            if ok and f3 < F0:                                     # ref :228-234
                X0 = list(newp)
                F0, dF0 = f3, list(df3)
            M -= 1                                                 # ref :235-236
            i += sb
            d3 = _dot(df3, s)                                      # ref :238-239
        if obj_flag and abs(d3) < -1.0 * SIG * d0 and f3 < opt_loss + x3 * RHO * d0:   # ref :242
            X = [X[j] + x3 * s[j] for j in range(len(X))]
            opt_loss = f3
            df3_df3, df3_df0, df0_df0 = _dot(df3, df3), _dot(df3, df0), _dot(df0, df0)   # ref :251-253
            s = [((df3_df3 - df3_df0) / df0_df0) * s[j] - df3[j] for j in range(len(s))]  # ref :254-256
            df0 = list(df3)                                        # ref :257
            d3 = d0
            d0 = _dot(df0, s)                                      # ref :259 (b aliases the re-assigned df0)
            if d0 > 0:                                             # ref :261-266
                s = [-1.0 * g for g in df0]
                d0 = -1.0 * _dot(s, s)
            x3 = x3 * min(RATIO, d3 / (d0 - 2.0 ** -52))           # ref :267
        else:                                                      # ref :270-281
            X, opt_loss, df0 = list(X0), F0, list(dF0)
            s = [-1.0 * g for g in df0]
            d0 = -1.0 * _dot(s, s)
            x3 = 1.0 / (1.0 - d0)
    return opt_loss, X, n_eval


class VarEMPrior:
    """The part of c_prior the variational-EM loop owns (ref: prior/c_prior.cpp:222-279, :109-116):
    cov_varEM = [psi: Q D R | delta: Q D R | phi: Q R | tau: Q R], all 1.0; cov_varEM_fix = [alpha, beta, gamma, d, eta]
    = [0.5, 0.5, 0.5, 0.5, eta] read as FLOAT by the optimiser (c_optimizer_varEM.cpp:42,98-102); per A entry the linked
    Normal prior (type 1, mean 0, variance psi stored as FLOAT in fix_param_cov, c_prior.h:46-48) or clamp (type 0)."""

    def __init__(self, Q, D, R, eta):
        self.Q, self.D, self.R = Q, D, R
        self.cov_varEM = [1.0] * (2 * Q * (D * R + R))
        self.fix = [np.float32(0.5), np.float32(0.5), np.float32(0.5), np.float32(0.5), np.float32(eta)]
        self.type_A = [1] * (Q * D * R)
        self.var_A = [np.float32(1.0)] * (Q * D * R)


def varem(max_iteration, init_parameter, obj_of_prior, prior, nlik, sub_opt_iter):
    """ref: c_optimizer_varEM.cpp:26-163.  obj_of_prior(prior) -> obj(theta) builds the objective for the current
    prior state (the reference's objective reads the c_prior object it shares with the optimiser).
    Returns (opt_loss, opt_parameter, list of per-iteration (loss, evaluations))."""
    Q, D, R = prior.Q, prior.D, prior.R
    opt_parameter = [float(v) for v in init_parameter]
    opt_loss, best_loss = float("nan"), float("nan")
    trace = []
    for it in range(abs(max_iteration)):                            # ref :59
        cur = 100 if it < 5 else sub_opt_iter                       # ref :62-68
        opt_loss, opt_parameter, nev = scg(-cur, opt_parameter, obj_of_prior(prior))   # ref :70-82
        trace.append((opt_loss, nev))
        if it > 0:                                                  # ref :88-95
            change_ratio = (opt_loss - best_loss) / best_loss
            if abs(change_ratio) < 0.005:
                break
        best_loss = opt_loss
        alpha, beta, gamma, dd, eta = prior.fix                     # ref :98-102 (floats)
        cv = prior.cov_varEM
        f32 = np.float32
        for q in range(Q):                                          # tau, ref :105-111, :165-174
            for r in range(R):
                index = Q * (2 * D * R + R) + q * R + r
                phi = cv[index - Q * R]
                cv[index] = float(f32(gamma + dd)) / (phi + float(eta))
        for q in range(Q):                                          # phi, ref :114-125, :175-186
            for r in range(R):
                index = Q * (2 * D * R) + q * R + r
                delta_sum = 0.0
                for d in range(D):
                    delta_sum += cv[Q * D * R + q * D * R + d * R + r]
                tau = cv[index + Q * R]
                cv[index] = (float(f32(f32(f32(D) * beta) + gamma)) - 1.0) / (delta_sum + tau)
        for q in range(Q):                                          # delta, ref :128-137, :187-197
            for d in range(D):
                for r in range(R):
                    index = Q * D * R + q * D * R + d * R + r
                    psi = cv[index - Q * D * R]
                    phi = cv[2 * Q * D * R + q * R + r]
                    cv[index] = float(f32(alpha + beta)) / (psi + phi)
        for q in range(Q):                                          # psi, ref :140-161, :198-208
            for d in range(D):
                for r in range(R):
                    index = q * D * R + d * R + r
                    a = opt_parameter[nlik + index]
                    delta = cv[index + Q * D * R]
                    sub = 2.0 * float(alpha) - 3.0
                    cv[index] = (sub + math.sqrt(sub * sub + 8.0 * delta * a * a)) / (4.0 * delta)
                    if cv[index] == 0.0:                            # ref :151-154
                        prior.type_A[index] = 0
                        opt_parameter[nlik + index] = 0.0
                    prior.var_A[index] = np.float32(cv[index])      # ref :157-158 (vector<float>)
    return opt_loss, opt_parameter, trace
