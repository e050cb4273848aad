// medgp_optimizer.hpp -- the reference's optimisers as RESUMABLE state machines, so that many patients can be
// advanced in lock step and every step issues ONE batched medgp_nlml_grad.
//
//   scg_machine    <- c_optimizer_scg::optimize   (ref: util/c_optimizer_scg.cpp:25-284): Rasmussen's minimize.m
//                     (Polak-Ribiere CG, cubic/quadratic line search; INT 0.1, EXT 3, MAX 20, RATIO 10, SIG 0.1,
//                     RHO 0.05, ref :37-42).  A negative max_iteration counts FUNCTION EVALUATIONS
//                     (i += signbit(max_iteration), ref :73,88,114,234).
//   varem_machine  <- c_optimizer_varEM::optimize (ref: util/c_optimizer_varEM.cpp:26-163): outer variational-EM
//                     loop (100 evaluations for the first 5 iterations, then iteration_num_per_update; 0.5 % early
//                     stop; closed-form tau / phi / delta / psi updates :165-206; psi becomes the Normal prior
//                     variance of its A entry; psi == 0 clamps the entry to 0).
// The single-patient classes c_optimizer_scg / c_optimizer_varEM keep the reference's optimize(...) signature and
// simply drive one machine against a c_objective_one.
#pragma once
#include <vector>

#include "medgp_host.hpp"

namespace medgp {

class scg_machine {
public:
    void start(int max_iteration, const std::vector<double> &init_parameter);
    bool done() const { return wait == W_DONE; }
    // the point whose (f, grad) the machine is waiting for; valid while !done()
    const std::vector<double> &request() const { return pending; }
    // result of evaluating request(): ok = the objective's bool, f / g untouched by the machine when !ok
    void feed(bool ok, double f, const std::vector<double> &g);

    double opt_loss = 0.0;
    std::vector<double> opt_parameter;
    int evaluations() const { return n_eval; }
    bool failed_init() const { return init_failed; }

private:
    enum Wait { W_INIT, W_EXTRAP, W_INTERP, W_DONE };
    enum Pc { P_OUTER_TOP, P_EXTRAP_TOP, P_EXTRAP_TRY, P_EXTRAP_POST, P_INTERP_TOP, P_LINE_END };
    void advance();
    void set_request(double step);
    Wait wait = W_DONE;
    Pc pc = P_OUTER_TOP;
    int max_it = 0, sb = 0, i = 0, n_eval = 0;
    bool ls_failed = false, obj_flag = false, success = false, init_failed = false;
    double M = 0, d0 = 0, x1 = 0, x2 = 0, x3 = 0, x4 = 0, d1 = 0, d2 = 0, d3 = 0, d4 = 0, f1 = 0, f2 = 0, f3 = 0, f4 = 0, F0 = 0;
    std::vector<double> s, df0, df3, dF0, X0, pending, init;
};

class varem_machine {
public:
    // max_iteration as passed by main_one_train (negative: (-1)*top_iteration_num); prior is edited in place
    void start(int max_iteration, const std::vector<double> &init_parameter, c_prior *prior, const std::vector<int> &kernel_param,
               int num_lik, int sub_opt_iter, bool display = true);
    bool done() const { return finished; }
    const std::vector<double> &request() const { return scg.request(); }
    void feed(bool ok, double f, const std::vector<double> &g);
    bool prior_changed() { bool c = prior_dirty; prior_dirty = false; return c; }

    double opt_loss = 0.0;
    std::vector<double> opt_parameter;

private:
    void begin_scg();
    void after_scg();
    scg_machine scg;
    c_prior *prior = nullptr;
    int max_it = 0, iter = 0, Q = 0, D = 0, R = 0, nlik = 0, sub_iter = 0;
    bool finished = true, prior_dirty = false, display = true;
    double best_loss = 0.0;
};

// reference-signature wrappers (single patient, one evaluation per call)
class c_optimizer_scg {
public:
    void optimize(const int &max_iteration, const std::vector<double> &init_parameter, c_objective_one *objfunc,
                  const bool &display, double &opt_loss, std::vector<double> &opt_parameter, c_kernel *&input_kernel,
                  c_meanfunc *&input_meanfunc, c_likelihood *&input_likfunc, c_inference_hip *&input_inffunc,
                  c_prior *&input_prior);
};
class c_optimizer_varEM {
public:
    void set_sub_opt_iter(const int &opt_iter) { sub_opt_iter = opt_iter; }
    void optimize(const int &max_iteration, const std::vector<double> &init_parameter, c_objective_one *objfunc,
                  const bool &display, double &opt_loss, std::vector<double> &opt_parameter, c_kernel *&input_kernel,
                  c_meanfunc *&input_meanfunc, c_likelihood *&input_likfunc, c_inference_hip *&input_inffunc,
                  c_prior *&input_prior);

private:
    int sub_opt_iter = 30;
};

}  // namespace medgp
