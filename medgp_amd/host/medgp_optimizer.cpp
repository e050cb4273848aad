// medgp_optimizer.cpp -- see medgp_optimizer.hpp.  "ref:" = /root/reference/medgpc/src/util/...
#include "medgp_optimizer.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <iostream>

namespace medgp {

namespace {
const double INT_ = 0.1, EXT_ = 3.0, MAX_ = 20, RATIO_ = 10, SIG_ = 0.1, RHO_ = SIG_ / 2.0;   // ref c_optimizer_scg.cpp:37-42
double dot(const std::vector<double> &a, const std::vector<double> &b) {   // cblas_ddot in the reference
    double s = 0.0;
    for (size_t k = 0; k < a.size(); k++) s += a[k] * b[k];
    return s;
}
}  // namespace

// ------------------------------------------------------------------------------------------- scg_machine
void scg_machine::start(int max_iteration, const std::vector<double> &init_parameter) {
    max_it = max_iteration;
    sb = std::signbit((double)max_iteration) ? 1 : 0;
    i = 0; n_eval = 0;
    ls_failed = false; init_failed = false;
    init = init_parameter;
    pending = init_parameter;
    opt_parameter = init_parameter;
    wait = W_INIT;
}

void scg_machine::set_request(double step) {
    pending.resize(opt_parameter.size());
    for (size_t j = 0; j < opt_parameter.size(); j++) pending[j] = opt_parameter[j] + step * s[j];
}

void scg_machine::feed(bool ok, double f, const std::vector<double> &g) {
    n_eval++;
    obj_flag = ok;
    switch (wait) {
    case W_INIT: {   // ref :64-84
        if (!ok) { init_failed = true; wait = W_DONE; opt_loss = f; return; }
        const double f0 = f;
        df0 = g;
        i = i + sb;
        s.resize(df0.size());
        for (size_t j = 0; j < df0.size(); j++) s[j] = -1.0 * df0[j];
        d0 = -1.0 * dot(s, s);
        x3 = 1.0 / (1.0 - d0);
        opt_loss = f0;
        opt_parameter = init;
        pc = P_OUTER_TOP;
        break;
    }
    case W_EXTRAP:   // ref :120-131
        if (ok) { f3 = f; df3 = g; }
        if (!ok || std::isinf(f3) || std::isnan(f3)) x3 = (x2 + x3) / 2.0;
        else success = true;
        pc = P_EXTRAP_TRY;
        break;
    case W_INTERP:   // ref :221-237
        if (ok) { f3 = f; df3 = g; }
        if (ok && f3 < F0) {
            for (size_t j = 0; j < opt_parameter.size(); j++) X0[j] = opt_parameter[j] + x3 * s[j];
            F0 = f3;
            dF0 = df3;
        }
        M = M - 1;
        i = i + sb;
        d3 = dot(df3, s);
        pc = P_INTERP_TOP;
        break;
    case W_DONE: return;
    }
    advance();
}

void scg_machine::advance() {
    while (true) {
        switch (pc) {
        case P_OUTER_TOP:   // ref :87-100
            if (!(i < std::abs(max_it))) { wait = W_DONE; return; }
            // a positive budget counts line searches as minimize.m does (the reference's port only ever adds
            // signbit(max_iteration), i.e. 0, and would never terminate; both mains pass negative budgets)
            i = i + sb + (max_it > 0 ? 1 : 0);
            X0 = opt_parameter;
            F0 = opt_loss;
            dF0 = df0;
            M = (max_it > 0) ? MAX_ : (double)std::min((int)MAX_, std::abs(max_it) - i);
            pc = P_EXTRAP_TOP;
            break;
        case P_EXTRAP_TOP:   // ref :102-111
            x2 = 0.0; f2 = opt_loss; d2 = d0;
            f3 = opt_loss; df3 = df0;
            success = false;
            pc = P_EXTRAP_TRY;
            break;
        case P_EXTRAP_TRY:   // ref :113-132
            if (!success && M > 0) {
                M = M - 1;
                i = i + sb;
                set_request(x3);
                wait = W_EXTRAP;
                return;
            }
            pc = P_EXTRAP_POST;
            break;
        case P_EXTRAP_POST: {   // ref :134-179
            if (f3 < F0) {
                for (size_t j = 0; j < opt_parameter.size(); j++) X0[j] = opt_parameter[j] + x3 * s[j];
                F0 = f3;
                dF0 = df3;
            }
            d3 = dot(df3, s);
            if ((d3 > SIG_ * d0) || (f3 > (opt_loss + x3 * RHO_ * d0)) || (M == 0)) { pc = P_INTERP_TOP; break; }
            x1 = x2; f1 = f2; d1 = d2;
            x2 = x3; f2 = f3; d2 = d3;
            const double A = 6.0 * (f1 - f2) + 3.0 * (d2 + d1) * (x2 - x1);
            const double B = 3.0 * (f2 - f1) - (2.0 * d1 + d2) * (x2 - x1);
            const double temp = B * B - A * d1 * (x2 - x1);
            if (temp < 0) x3 = x2 * EXT_;
            else {
                x3 = x1 - (d1 * std::pow(x2 - x1, 2.0) / (B + std::sqrt(temp)));
                if (std::isnan(x3) || std::isinf(x3) || (x3 < 0)) x3 = x2 * EXT_;
                else if (x3 > x2 * EXT_) x3 = x2 * EXT_;
                else if (x3 < (x2 + INT_ * (x2 - x1))) x3 = x2 + INT_ * (x2 - x1);
            }
            pc = P_EXTRAP_TOP;
            break;
        }
        case P_INTERP_TOP: {   // ref :181-219
            if (((std::fabs(d3) > -1.0 * SIG_ * d0) || (f3 > (opt_loss + x3 * RHO_ * d0))) && (M > 0)) {
                if ((d3 > 0) || (f3 > (opt_loss + x3 * RHO_ * d0))) { x4 = x3; f4 = f3; d4 = d3; }
                else { x2 = x3; f2 = f3; d2 = d3; }
                if (f4 > opt_loss) {
                    x3 = x2 - (0.5 * d2 * std::pow(x4 - x2, 2.0)) / (f4 - f2 - d2 * (x4 - x2));
                    if (std::isnan(x3) || std::isinf(x3)) x3 = (x2 + x4) / 2.0;
                } else {
                    const double A = 6.0 * (f2 - f4) / (x4 - x2) + 3.0 * (d4 + d2);
                    const double B = 3.0 * (f4 - f2) - (2.0 * d2 + d4) * (x4 - x2);
                    const double disc = B * B - A * d2 * std::pow(x4 - x2, 2.0);
                    if (disc < 0) x3 = (x2 + x4) / 2.0;
                    else {
                        x3 = x2 + (std::sqrt(disc) - B) / A;
                        if (std::isnan(x3) || std::isinf(x3)) x3 = (x2 + x4) / 2.0;
                    }
                }
                x3 = std::max(std::min(x3, x4 - INT_ * (x4 - x2)), x2 + INT_ * (x4 - x2));
                set_request(x3);
                wait = W_INTERP;
                return;
            }
            pc = P_LINE_END;
            break;
        }
        case P_LINE_END:   // ref :240-281
            if (obj_flag && (std::fabs(d3) < -1.0 * SIG_ * d0) && (f3 < (opt_loss + x3 * RHO_ * d0))) {
                for (size_t j = 0; j < opt_parameter.size(); j++) opt_parameter[j] = opt_parameter[j] + x3 * s[j];
                opt_loss = f3;
                const double df3_df3 = dot(df3, df3), df3_df0 = dot(df3, df0), df0_df0 = dot(df0, df0);
                for (size_t j = 0; j < df3.size(); j++) s[j] = ((df3_df3 - df3_df0) / df0_df0) * s[j] - df3[j];
                df0 = df3;
                d3 = d0;
                d0 = dot(df0, s);
                if (d0 > 0) {
                    for (size_t j = 0; j < df0.size(); j++) s[j] = -1.0 * df0[j];
                    d0 = -1.0 * dot(s, s);
                }
                x3 = x3 * std::min(RATIO_, d3 / (d0 - std::pow(2.0, -52)));
                ls_failed = false;
            } else {
                opt_parameter = X0;
                opt_loss = F0;
                df0 = dF0;
                for (size_t j = 0; j < df0.size(); j++) s[j] = -1.0 * df0[j];
                d0 = -1.0 * dot(s, s);
                x3 = 1.0 / (1.0 - d0);
                ls_failed = true;
            }
            pc = P_OUTER_TOP;
            break;
        }
    }
}

// ------------------------------------------------------------------------------------------- varem_machine
void varem_machine::start(int max_iteration, const std::vector<double> &init_parameter, c_prior *p,
                          const std::vector<int> &kernel_param, int num_lik, int sub_opt_iter, bool disp) {
    max_it = max_iteration; prior = p; nlik = num_lik; sub_iter = sub_opt_iter; display = disp;
    Q = kernel_param[0]; D = kernel_param[1]; R = kernel_param[2];
    opt_parameter = init_parameter;
    iter = 0;
    finished = false;
    prior_dirty = false;
    if (iter < std::abs(max_it)) begin_scg();
    else finished = true;
}

void varem_machine::begin_scg() {
    const int curr = (iter < 5) ? 100 : sub_iter;   // ref c_optimizer_varEM.cpp:64-70
    scg.start(-curr, opt_parameter);
}

void varem_machine::feed(bool ok, double f, const std::vector<double> &g) {
    scg.feed(ok, f, g);
    while (!finished && scg.done()) after_scg();
}

void varem_machine::after_scg() {
    opt_loss = scg.opt_loss;
    opt_parameter = scg.opt_parameter;
    if (display) std::cout << "iteration " << iter << " for variational EM: loss = " << opt_loss << std::endl;
    if (iter > 0) {   // ref :89-95
        const double change_ratio = (opt_loss - best_loss) / best_loss;
        if (std::abs(change_ratio) < 0.005) {
            std::cout << "change of loss " << change_ratio << " meets early stop criterion" << std::endl;
            finished = true;
            return;
        }
    }
    best_loss = opt_loss;
    // closed-form updates (ref :98-162, :165-206); the fixed parameters are floats in the reference
    const float alpha = (float)prior->get_cov_varEM_fix_one(0), beta = (float)prior->get_cov_varEM_fix_one(1),
                gamma = (float)prior->get_cov_varEM_fix_one(2), dd = (float)prior->get_cov_varEM_fix_one(3),
                eta = (float)prior->get_cov_varEM_fix_one(4);
    for (int q = 0; q < Q; q++)
        for (int r = 0; r < R; r++) {   // tau
            const int index = Q * (2 * D * R + R) + q * R + r;
            const double phi = prior->get_cov_varEM_one(index - Q * R);
            prior->set_cov_varEM_one((gamma + dd) / (phi + eta), index);
        }
    for (int q = 0; q < Q; q++)
        for (int r = 0; r < R; r++) {   // phi
            const int index = Q * (2 * D * R) + q * R + r;
            double delta_sum = 0.0;
            for (int d = 0; d < D; d++) delta_sum += prior->get_cov_varEM_one(Q * D * R + q * D * R + d * R + r);
            const double tau = prior->get_cov_varEM_one(index + Q * R);
            prior->set_cov_varEM_one((((float)D) * beta + gamma - 1.0) / (delta_sum + tau), index);
        }
    for (int q = 0; q < Q; q++)
        for (int d = 0; d < D; d++)
            for (int r = 0; r < R; r++) {   // delta
                const int index = Q * D * R + q * D * R + d * R + r;
                const double psi = prior->get_cov_varEM_one(index - Q * D * R);
                const double phi = prior->get_cov_varEM_one(2 * Q * D * R + q * R + r);
                prior->set_cov_varEM_one((alpha + beta) / (psi + phi), index);
            }
    const int offset = nlik;
    for (int q = 0; q < Q; q++)
        for (int d = 0; d < D; d++)
            for (int r = 0; r < R; r++) {   // psi, and the linked Normal prior of the A entry
                const int index = q * D * R + d * R + r;
                const double a = opt_parameter[offset + index];
                const double delta = prior->get_cov_varEM_one(index + Q * D * R);
                const double sub = (2.0 * alpha - 3.0);
                double new_psi = sub + std::sqrt(sub * sub + 8.0 * delta * a * a);
                new_psi = new_psi / (4.0 * delta);
                prior->set_cov_varEM_one(new_psi, index);
                if (prior->get_cov_varEM_one(index) == 0.0) {
                    prior->type_cov[index] = 0;
                    opt_parameter[offset + index] = 0.0;
                }
                prior->fix_param_cov[index][0] = 0;
                prior->fix_param_cov[index][1] = (float)prior->get_cov_varEM_one(index);
            }
    prior->touch();
    prior_dirty = true;
    iter++;
    if (iter < std::abs(max_it)) begin_scg();
    else finished = true;
}

// ------------------------------------------------------------------------------------------- reference-signature wrappers
void c_optimizer_scg::optimize(const int &max_iteration, const std::vector<double> &init_parameter, c_objective_one *objfunc,
                               const bool &display, double &opt_loss, std::vector<double> &opt_parameter, c_kernel *&k,
                               c_meanfunc *&m, c_likelihood *&l, c_inference_hip *&inf, c_prior *&p) {
    scg_machine mc;
    mc.start(max_iteration, init_parameter);
    double f = 0.0;
    std::vector<double> g;
    while (!mc.done()) {
        const std::vector<double> th = mc.request();
        bool ok = objfunc->compute_objective(true, th, f, g, k, m, l, inf, p);
        mc.feed(ok, f, g);
        if (display) std::cout << (max_iteration > 0 ? "Linesearch " : "Function evaluation ") << mc.evaluations() << ": " << mc.opt_loss << std::endl;
    }
    opt_loss = mc.opt_loss;
    opt_parameter = mc.opt_parameter;
}

void c_optimizer_varEM::optimize(const int &max_iteration, const std::vector<double> &init_parameter, c_objective_one *objfunc,
                                 const bool &display, double &opt_loss, std::vector<double> &opt_parameter, c_kernel *&k,
                                 c_meanfunc *&m, c_likelihood *&l, c_inference_hip *&inf, c_prior *&p) {
    const std::vector<int> kp = k->get_kernel_param();
    if ((int)kp.size() != 3) { std::cout << "ERROR: varEM is only usable for LMCSM kernel!" << std::endl; return; }
    varem_machine mc;
    mc.start(max_iteration, init_parameter, p, kp, l->get_likfunc_hyp_num(), sub_opt_iter, display);
    double f = 0.0;
    std::vector<double> g;
    while (!mc.done()) {
        const std::vector<double> th = mc.request();
        bool ok = objfunc->compute_objective(true, th, f, g, k, m, l, inf, p);
        mc.feed(ok, f, g);
    }
    opt_loss = mc.opt_loss;
    opt_parameter = mc.opt_parameter;
}

}  // namespace medgp
