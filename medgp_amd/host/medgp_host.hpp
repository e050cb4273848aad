// medgp_host.hpp -- C++ host side above the C ABI (include/medgp_hip.h), mirroring the reference's
// plug-in interface for the hot path: same class names, method names, argument meaning and error
// behaviour (bool returns, no exceptions across the evaluation path), so that a host written like
// main_one_train.cpp / main_one_test.cpp switches by changing the inference object.
//
//   reference                                   here (namespace medgp)
//   c_kernel_LMC_SM / _SE / _SM  (kernel/*.h)   c_kernel            -- parameter / hyper holder only (no math on host)
//   c_likelihood_gaussian(MO)                   c_likelihood        -- holder
//   c_meanfunc_zero                             c_meanfunc_zero     -- holder (the only mean either main constructs)
//   c_prior                      (prior/c_prior.h)   c_prior        -- same public vectors + setup/init methods
//   c_inference_prior / _exact   (inference/*.h)     c_inference_hip-- compute_nlml(...) with the reference signature
//   GP_Regression                (core/gp_regression.h) GP_Regression -- train / predict
//   c_objective_one              (util/c_objective_one.h) c_objective_one -- compute_objective(...)
//   c_hyperparam                 (core/c_hyperparam.h)    c_hyperparam
//
// "ref:" = /root/reference/medgpc/src/...
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/medgp_hip.h"

namespace medgp {

using std::vector;

// ref: core/c_hyperparam.h / c_hyperparam.cpp:99-122 -- theta = [lik | cov | mean]
class c_hyperparam {
public:
    c_hyperparam() {}
    c_hyperparam(const vector<double> &hyp_all, const int &num_cov, const int &num_mean, const int &num_lik) {
        set_hyp_all(hyp_all, num_cov, num_mean, num_lik);
    }
    void set_hyp_all(const vector<double> &input_hyp, const int &num_cov, const int &num_mean, const int &num_lik);
    vector<double> get_hyp_cov() const { return hyp_cov; }
    vector<double> get_hyp_mean() const { return hyp_mean; }
    vector<double> get_hyp_lik() const { return hyp_lik; }
    vector<double> get_hyp_all() const;

private:
    vector<double> hyp_cov, hyp_mean, hyp_lik;
};

// Covariance-family holder.  kernel_index 7: param = {Q, D, R}; 8: {Q}; 0: {}.
// ref: kernel/c_kernel.h:15-105, c_kernel_LMC_SM.cpp:51-70 (set_kernel_hyp exps indices >= Q*D*R),
//      c_kernel_SM.cpp / c_kernel_SE.cpp (exp on everything)
class c_kernel {
public:
    c_kernel(int kernel_index, const vector<int> &input_param);
    void set_kernel_hyp(const vector<double> &input_hyp);       // raw (optimiser) values
    vector<double> get_kernel_hyp() const { return kernel_hyp; } // transformed, like the reference
    vector<double> get_kernel_hyp_raw() const { return kernel_hyp_raw; }
    vector<int> get_kernel_param() const { return kernel_param; }
    int get_kernel_hyp_num() const { return kernel_hyp_num; }
    int get_kernel_index() const { return kernel_index; }
    int Q() const, D() const, R() const;

private:
    int kernel_index, kernel_hyp_num;
    vector<int> kernel_param;
    vector<double> kernel_hyp, kernel_hyp_raw;
};

// ref: likelihoods/c_likelihood.h, c_likelihood.cpp:38-43 (exp), gaussianMO: D hypers, gaussian: 1
class c_likelihood {
public:
    explicit c_likelihood(int num_hyp) : likfunc_hyp_num(num_hyp) {}
    void set_likfunc_hyp(const vector<double> &input_hyp);
    vector<double> get_likfunc_hyp() const { return likfunc_hyp; }
    vector<double> get_likfunc_hyp_raw() const { return likfunc_hyp_raw; }
    int get_likfunc_hyp_num() const { return likfunc_hyp_num; }

private:
    int likfunc_hyp_num;
    vector<double> likfunc_hyp, likfunc_hyp_raw;
};

// ref: mean/c_meanfunc_zero.h -- no hypers, mean 0
class c_meanfunc_zero {
public:
    void set_meanfunc_hyp(const vector<double> &) {}
    vector<double> get_meanfunc_hyp() const { return {}; }
    int get_meanfunc_hyp_num() const { return 0; }
};
typedef c_meanfunc_zero c_meanfunc;

// ref: prior/c_prior.h:12-98, prior/c_prior.cpp
class c_prior {
public:
    c_prior() {}
    c_prior(int num_cov, int num_mean, int num_lik) { initialize_param(num_cov, num_mean, num_lik); }
    void initialize_param(int num_cov, int num_mean, int num_lik);                                   // ref :59-107
    void setup_param(const int kernel_index, const vector<int> &kernel_param, const int &mode,
                     const vector<float> &prior_param);                                              // ref :196-220
    void setup_hier_gamma_prior(const vector<int> &kernel_param, const vector<float> &prior_param);  // ref :222-279
    void init_test_prior(const int kernel_index, const vector<int> &test_kernel_param,
                         const vector<double> &test_mode_param);                                     // ref :118-140

    vector<bool> flag_cov, flag_mean, flag_lik;   // prior active
    vector<bool> exp_cov, exp_mean, exp_lik;      // hyper uses the exp transform (chain rule)
    vector<vector<float>> fix_param_cov, fix_param_mean, fix_param_lik;
    vector<int> type_cov, type_mean, type_lik;    // -1 none, 0 clamp, 1 normal, 2 laplace

    void set_cov_varEM_all(const vector<double> &v) { cov_varEM = v; }
    void set_cov_varEM_one(double value, const int &index) { cov_varEM[index] = value; }
    vector<double> get_cov_varEM_all() const { return cov_varEM; }
    double get_cov_varEM_one(const int &index) const { return cov_varEM[index]; }
    void set_cov_varEM_fix_one(double value, const int &index) { cov_varEM_fix[index] = value; }
    double get_cov_varEM_fix_one(const int &index) const { return cov_varEM_fix[index]; }
    bool get_one_prior_flag(const int &index) const;   // index over [lik | cov | mean], ref :141-151
    int get_one_prior_type(const int &index) const;    // ref :153-163

    // flat per-hyper arrays in theta order [lik | cov | mean] for medgp_set_prior
    void flatten(vector<uint8_t> &flag, vector<int32_t> &type, vector<uint8_t> &is_exp, vector<float> &p0,
                 vector<float> &p1) const;
    uint64_t version() const { return version_; }
    void touch() { version_++; }   // call after editing the public vectors in place (varEM does)

private:
    int hyp_cov_num = 0, hyp_mean_num = 0, hyp_lik_num = 0;
    vector<double> cov_varEM_fix, cov_varEM;
    uint64_t version_ = 1;
};

// The operator.  Owns one medgp_ctx (device, family); not re-entrant, like the reference's objects.
// ref: inference/c_inference.h:38-52, c_inference_prior.cpp:25-154, c_inference_exact.cpp:29-244
class c_inference_hip {
public:
    explicit c_inference_hip(const int &thread_num = 1, int device = 0);   // thread_num kept for signature parity; unused
    ~c_inference_hip();
    c_inference_hip(const c_inference_hip &) = delete;
    c_inference_hip &operator=(const c_inference_hip &) = delete;

    int get_thread_num() const { return inf_thread_num; }
    const std::string &last_error() const { return err; }

    // Same argument list as c_inference::compute_nlml.  chol_alpha / chol_factor_inv are caller-allocated
    // N*N float buffers exactly as GP_Regression::train allocates them (ref core/gp_regression.cpp:111-117);
    // either may be NULL to skip the copy-out.  Returns false on Cholesky failure after 10 jitters.
    bool compute_nlml(const bool &flag_grad, const vector<int> &meta, const vector<float> &x, const vector<float> &y,
                      c_kernel *kernel, c_meanfunc *meanfunc, c_likelihood *likfunc, c_prior *prior, float *&chol_alpha,
                      float *&chol_factor_inv, float &beta, double &nlml, vector<double> &dnlml);

    // predictive mean / variance for the LAST patient passed to compute_nlml (same theta is re-sent)
    bool predict(c_kernel *kernel, c_likelihood *likfunc, const vector<int> &meta2, const vector<float> &x2,
                 vector<float> &mean, vector<float> &var);

    int last_status() const { return status; }   // jitter rounds (0..10) or -1
    medgp_ctx *ctx() { return ctx_; }

private:
    bool ensure(c_kernel *kernel, int n);
    bool upload(const vector<int> &meta, const vector<float> &x, const vector<float> &y);
    bool sync_prior(c_kernel *kernel, c_likelihood *likfunc, c_prior *prior);
    int inf_thread_num, device_;
    medgp_ctx *ctx_ = nullptr;
    int kidx = -1, Q = 0, D = 0, R = 0, cap_n = 0;
    vector<int> cur_meta;
    vector<float> cur_x, cur_y;
    bool have_patient = false;
    const c_prior *prior_seen = nullptr;
    uint64_t prior_version = 0;
    int status = 0;
    std::string err;
};

// ref: core/gp_regression.h / gp_regression.cpp:25-214
class GP_Regression {
public:
    GP_Regression(const int &input_dim, c_kernel *input_kernel, c_meanfunc *input_meanfunc, c_likelihood *input_likfunc,
                  c_inference_hip *input_inffunc, c_prior *input_prior);
    ~GP_Regression();
    bool get_flag_trained() const { return flag_trained; }
    double get_neg_log_mlikelihood() const { return nlm_likelihood; }
    vector<double> get_dneg_log_mlikelihood() const { return dnlm_likelihood; }
    void train(const bool &flag_grad, const vector<int> &meta, const vector<float> &x, const vector<float> &y);
    // returns {mean[N*], var[N*]} like the reference
    vector<vector<float>> predict(const vector<int> &meta, const vector<int> &meta2, const vector<float> &x,
                                  const vector<float> &y, const vector<float> &x2);
    const float *get_chol_alpha() const { return chol_alpha; }
    const float *get_chol_factor_inv() const { return chol_factor_inv; }
    float get_beta() const { return beta; }

private:
    int dim;
    bool flag_trained = false;
    double nlm_likelihood = 0.0;
    vector<double> dnlm_likelihood;
    float *chol_factor_inv = nullptr, *chol_alpha = nullptr;
    float beta = 0.f;
    c_kernel *kernel;
    c_meanfunc *meanfunc;
    c_likelihood *likfunc;
    c_inference_hip *inffunc;
    c_prior *prior;
};

// ref: util/c_objective.h:29-39, util/c_objective_one.h / c_objective_one.cpp:40-82
class c_objective_one {
public:
    c_objective_one(const int &kernel_idx, const vector<int> &kernel_param, const vector<int> &meta,
                    const vector<float> &x, const vector<float> &y)
        : obj_meta(meta), obj_x(x), obj_y(y), obj_kernel_idx(kernel_idx), obj_kernel_param(kernel_param) {}
    bool compute_objective(const bool &flag_grad, const vector<double> &input_parameter, double &objective_value,
                           vector<double> &gradients, c_kernel *&input_kernel, c_meanfunc *&input_meanfunc,
                           c_likelihood *&input_likfunc, c_inference_hip *&input_inffunc, c_prior *&input_prior);
    const vector<int> &meta() const { return obj_meta; }
    const vector<float> &x() const { return obj_x; }
    const vector<float> &y() const { return obj_y; }

private:
    vector<int> obj_meta;
    vector<float> obj_x, obj_y;
    int obj_kernel_idx;
    vector<int> obj_kernel_param;
};

}  // namespace medgp
