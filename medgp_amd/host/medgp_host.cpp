// medgp_host.cpp -- see medgp_host.hpp.  Host logic only; all arithmetic of the hot path runs in
// libmedgp_hip.so.  "ref:" = /root/reference/medgpc/src/...
#include "medgp_host.hpp"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <iostream>

namespace medgp {

// ------------------------------------------------------------------------------------------- c_hyperparam
void c_hyperparam::set_hyp_all(const vector<double> &input_hyp, const int &num_cov, const int &num_mean, const int &num_lik) {
    hyp_lik.assign(input_hyp.begin(), input_hyp.begin() + num_lik);
    hyp_cov.assign(input_hyp.begin() + num_lik, input_hyp.begin() + num_lik + num_cov);
    hyp_mean.assign(input_hyp.begin() + num_lik + num_cov, input_hyp.begin() + num_lik + num_cov + num_mean);
}
vector<double> c_hyperparam::get_hyp_all() const {
    vector<double> a(hyp_lik);
    a.insert(a.end(), hyp_cov.begin(), hyp_cov.end());
    a.insert(a.end(), hyp_mean.begin(), hyp_mean.end());
    return a;
}

// ------------------------------------------------------------------------------------------- c_kernel
c_kernel::c_kernel(int kidx, const vector<int> &input_param) : kernel_index(kidx), kernel_param(input_param) {
    if (kidx == MEDGP_KERNEL_LMC_SM) {
        if (input_param.size() != 3) {   // ref: c_kernel_LMC_SM.cpp:38-42
            std::cout << "ERROR:current input parameters should report 3 numbers (mixture, output, rank); received "
                      << input_param.size() << std::endl;
            kernel_hyp_num = -1;
            return;
        }
        kernel_hyp_num = Q() * (D() * R() + 2 + D());   // ref: c_kernel_LMC_SM.cpp:64-70
    } else if (kidx == MEDGP_KERNEL_SM) {
        kernel_hyp_num = 3 * Q();
    } else if (kidx == MEDGP_KERNEL_SE) {
        kernel_hyp_num = 2;
    } else {
        std::cout << "Error: not supported kernel type " << kidx << std::endl;   // ref: main_one_train.cpp:94-97
        kernel_hyp_num = -1;
    }
}
int c_kernel::Q() const { return kernel_index == MEDGP_KERNEL_SE ? 1 : (kernel_param.empty() ? 1 : kernel_param[0]); }
int c_kernel::D() const { return kernel_index == MEDGP_KERNEL_LMC_SM ? kernel_param[1] : 1; }
int c_kernel::R() const { return kernel_index == MEDGP_KERNEL_LMC_SM ? kernel_param[2] : 0; }
void c_kernel::set_kernel_hyp(const vector<double> &input_hyp) {
    kernel_hyp_raw = input_hyp;
    kernel_hyp = input_hyp;
    const int first_exp = (kernel_index == MEDGP_KERNEL_LMC_SM) ? Q() * D() * R() : 0;   // ref: c_kernel_LMC_SM.cpp:57-59
    for (int i = first_exp; i < (int)input_hyp.size(); i++) kernel_hyp[i] = std::exp(kernel_hyp[i]);
}

void c_likelihood::set_likfunc_hyp(const vector<double> &input_hyp) {   // ref: c_likelihood.cpp:38-43
    likfunc_hyp_raw = input_hyp;
    likfunc_hyp = input_hyp;
    for (auto &v : likfunc_hyp) v = std::exp(v);
}

// ------------------------------------------------------------------------------------------- c_prior
void c_prior::initialize_param(int num_cov, int num_mean, int num_lik) {
    hyp_cov_num = num_cov; hyp_mean_num = num_mean; hyp_lik_num = num_lik;
    auto init = [](int n, vector<bool> &flag, vector<bool> &ex, vector<vector<float>> &fix, vector<int> &type) {
        flag.assign(n, false); ex.assign(n, false); fix.assign(n, vector<float>()); type.assign(n, -1);
    };
    init(num_cov, flag_cov, exp_cov, fix_param_cov, type_cov);
    init(num_mean, flag_mean, exp_mean, fix_param_mean, type_mean);
    init(num_lik, flag_lik, exp_lik, fix_param_lik, type_lik);
    cov_varEM.clear();
    cov_varEM_fix.clear();
    touch();
}

void c_prior::setup_param(const int kernel_index, const vector<int> &kernel_param, const int &mode,
                          const vector<float> &prior_param) {
    if (kernel_index != 7) {
        std::cout << "Warning: prior mode is only available for LMCSM kernel now; prior will not be effective" << std::endl;
        return;
    }
    switch (mode) {
    case 0: std::cout << "mode " << mode << ": no regularization" << std::endl; break;
    case 2:
        std::cout << "mode " << mode << ": setup hierarchical gamma prior" << std::endl;
        setup_hier_gamma_prior(kernel_param, prior_param);
        break;
    default: std::cout << "undefined setup mode " << mode << "; no changes" << std::endl; break;
    }
}

void c_prior::setup_hier_gamma_prior(const vector<int> &kernel_param, const vector<float> &prior_param) {
    const int Q = kernel_param[0], D = kernel_param[1], R = kernel_param[2];
    // variational-EM state: [psi: QDR | delta: QDR | phi: QR | tau: QR] all 1; fixed [alpha beta gamma d eta]
    cov_varEM.assign(2 * Q * (D * R + R), 1.0);
    cov_varEM_fix.assign(5, 0.5);
    cov_varEM_fix[4] = prior_param.size() > 0 ? prior_param[0] : 50.0;
    for (int i = 0; i < hyp_cov_num; i++) {
        if (i < Q * D * R) {                       // A ~ Normal(0, 1)
            flag_cov[i] = true; exp_cov[i] = false; fix_param_cov[i] = {0.0f, 1.0f}; type_cov[i] = 1;
        } else if (i < Q * (D * R + 2)) {          // mu, v: no prior, exp transform
            flag_cov[i] = false; exp_cov[i] = true;
        } else if (i < Q * (D * R + 2 + D)) {      // kappa ~ Laplace(0, beta_lam), exp transform
            flag_cov[i] = true; exp_cov[i] = true;
            fix_param_cov[i] = {0.0f, prior_param.size() > 1 ? prior_param[1] : 0.5f};
            type_cov[i] = 2;
        } else {
            flag_cov[i] = false; exp_cov[i] = true;
        }
    }
    touch();
}

void c_prior::init_test_prior(const int kernel_index, const vector<int> &p, const vector<double> &mode_param) {
    if (kernel_index != 7) {
        std::cout << "Warning: testing prior is only set for LMCSM kernel now;prior will not be effective" << std::endl;
        return;
    }
    std::cout << "Info: setup prior to fix zero A elements" << std::endl;
    const int Q = p[0], D = p[1], R = p[2];
    for (int i = D; i < D + Q * D * R; i++)
        if (mode_param[i] == 0.0) { flag_cov[i - D] = true; type_cov[i - D] = 0; }
    touch();
}

bool c_prior::get_one_prior_flag(const int &index) const {
    if (index < hyp_lik_num) return flag_lik[index];
    if (index < hyp_lik_num + hyp_cov_num) return flag_cov[index - hyp_lik_num];
    return flag_mean[index - hyp_lik_num - hyp_cov_num];
}
int c_prior::get_one_prior_type(const int &index) const {
    if (index < hyp_lik_num) return type_lik[index];
    if (index < hyp_lik_num + hyp_cov_num) return type_cov[index - hyp_lik_num];
    return type_mean[index - hyp_lik_num - hyp_cov_num];
}

void c_prior::flatten(vector<uint8_t> &flag, vector<int32_t> &type, vector<uint8_t> &is_exp, vector<float> &p0,
                      vector<float> &p1) const {
    flag.clear(); type.clear(); is_exp.clear(); p0.clear(); p1.clear();
    auto add = [&](const vector<bool> &f, const vector<bool> &e, const vector<vector<float>> &fx, const vector<int> &t) {
        for (size_t i = 0; i < f.size(); i++) {
            flag.push_back(f[i] ? 1 : 0);
            type.push_back(t[i]);
            is_exp.push_back(e[i] ? 1 : 0);
            p0.push_back(fx[i].size() > 0 ? fx[i][0] : 0.0f);
            p1.push_back(fx[i].size() > 1 ? fx[i][1] : 1.0f);
        }
    };
    add(flag_lik, exp_lik, fix_param_lik, type_lik);   // theta order: lik, cov, mean (ref c_hyperparam.cpp:99-122)
    add(flag_cov, exp_cov, fix_param_cov, type_cov);
    add(flag_mean, exp_mean, fix_param_mean, type_mean);
}

// ------------------------------------------------------------------------------------------- c_inference_hip
c_inference_hip::c_inference_hip(const int &thread_num, int device) : inf_thread_num(thread_num), device_(device) {}
c_inference_hip::~c_inference_hip() { if (ctx_) medgp_destroy(ctx_); }

bool c_inference_hip::ensure(c_kernel *kernel, int n) {
    const int k = kernel->get_kernel_index(), q = kernel->Q(), d = kernel->D(), r = kernel->R();
    if (ctx_ && (k != kidx || q != Q || d != D || r != R)) { medgp_destroy(ctx_); ctx_ = nullptr; }
    if (!ctx_) {
        int rc = medgp_create(&ctx_, device_, k, q, d, r);
        if (rc) { err = medgp_last_error(nullptr); std::cout << "ERROR: " << err << std::endl; ctx_ = nullptr; return false; }
        kidx = k; Q = q; D = d; R = r; cap_n = 0; have_patient = false; prior_seen = nullptr;
    }
    if (n > cap_n) {
        int want = std::max(n, std::max(64, cap_n * 2));
        if (medgp_reserve(ctx_, 1, want, 1)) { err = medgp_last_error(ctx_); std::cout << "ERROR: " << err << std::endl; return false; }
        cap_n = want; have_patient = false; prior_seen = nullptr;
    }
    return true;
}

bool c_inference_hip::upload(const vector<int> &meta, const vector<float> &x, const vector<float> &y) {
    if (have_patient && meta == cur_meta && x == cur_x && y == cur_y) return true;   // optimiser loops re-send the same data
    const int n = (int)y.size();
    const int32_t *mp = meta.empty() ? nullptr : (const int32_t *)meta.data();
    if (medgp_set_patient(ctx_, 0, n, mp, x.data(), y.data())) { err = medgp_last_error(ctx_); std::cout << "ERROR: " << err << std::endl; return false; }
    cur_meta = meta; cur_x = x; cur_y = y; have_patient = true;
    return true;
}

bool c_inference_hip::sync_prior(c_kernel *, c_likelihood *, c_prior *prior) {
    if (prior == nullptr) {
        if (prior_seen != nullptr || prior_version != 0) { medgp_set_prior(ctx_, 0, nullptr, nullptr, nullptr, nullptr, nullptr); }
        prior_seen = nullptr; prior_version = 0;
        return true;
    }
    if (prior == prior_seen && prior->version() == prior_version) return true;
    vector<uint8_t> flag, ex; vector<int32_t> type; vector<float> p0, p1;
    prior->flatten(flag, type, ex, p0, p1);
    if ((int)flag.size() != medgp_num_hyp(ctx_)) { err = "prior size does not match the number of hypers"; std::cout << "ERROR: " << err << std::endl; return false; }
    if (medgp_set_prior(ctx_, 0, flag.data(), type.data(), ex.data(), p0.data(), p1.data())) { err = medgp_last_error(ctx_); std::cout << "ERROR: " << err << std::endl; return false; }
    prior_seen = prior; prior_version = prior->version();
    return true;
}

bool c_inference_hip::compute_nlml(const bool &flag_grad, const vector<int> &meta, const vector<float> &x,
                                   const vector<float> &y, c_kernel *kernel, c_meanfunc *, c_likelihood *likfunc,
                                   c_prior *prior, float *&chol_alpha, float *&chol_factor_inv, float &beta, double &nlml,
                                   vector<double> &dnlml) {
    const int n = (int)y.size();
    if (!ensure(kernel, n) || !upload(meta, x, y) || !sync_prior(kernel, likfunc, prior)) return false;
    // theta = [lik | cov] raw values, as the optimiser holds them (ref: c_objective_one.cpp:54-62)
    vector<double> theta = likfunc->get_likfunc_hyp_raw();
    const vector<double> cov = kernel->get_kernel_hyp_raw();
    theta.insert(theta.end(), cov.begin(), cov.end());
    const int H = medgp_num_hyp(ctx_);
    if ((int)theta.size() != H) { err = "mismatch # of hyperparameters"; std::cout << "ERROR: " << err << std::endl; return false; }
    int32_t slot = 0, st = -1;
    vector<double> g(flag_grad ? H : 0);
    double f = 0.0;
    // the factor outputs (chol_alpha / chol_factor_inv / beta) are formed only when the caller passes buffers for them
    const bool want_factor = chol_alpha || chol_factor_inv;
    const int flags = (flag_grad ? MEDGP_FLAG_GRAD : 0) | (want_factor ? MEDGP_FLAG_KEEP_FACTOR : 0);
    if (medgp_nlml_grad(ctx_, 1, &slot, theta.data(), flags, &f, flag_grad ? g.data() : nullptr, &st)) {
        err = medgp_last_error(ctx_); std::cout << "ERROR: " << err << std::endl; return false;
    }
    status = st;
    if (st < 0) return false;   // the reference's `return flag_success` = false
    if (st > 0) std::cout << "WARNING: Cholesky decomposition failed! jitter rounds = " << st << std::endl;
    nlml = f;
    if (flag_grad) dnlml = g;   // ref: dnlml cleared + filled in theta order (c_inference_exact.cpp:158-160)
    if (want_factor) {
        float b = 0.f;
        if (medgp_get_factor(ctx_, 0, chol_alpha, chol_factor_inv, &b)) { err = medgp_last_error(ctx_); std::cout << "ERROR: " << err << std::endl; return false; }
        beta = b;
    }
    return true;
}

bool c_inference_hip::predict(c_kernel *kernel, c_likelihood *likfunc, const vector<int> &meta2, const vector<float> &x2,
                              vector<float> &mean, vector<float> &var) {
    if (!ctx_ || !have_patient) return false;
    vector<double> theta = likfunc->get_likfunc_hyp_raw();
    const vector<double> cov = kernel->get_kernel_hyp_raw();
    theta.insert(theta.end(), cov.begin(), cov.end());
    const int ns = (int)x2.size();
    mean.assign(ns, 0.f); var.assign(ns, 0.f);
    int32_t st = -1;
    const int32_t *mp = meta2.empty() ? nullptr : (const int32_t *)meta2.data();
    if (medgp_fit_predict(ctx_, 0, theta.data(), ns, mp, x2.data(), mean.data(), var.data(), &st)) {
        err = medgp_last_error(ctx_); std::cout << "ERROR: " << err << std::endl; return false;
    }
    status = st;
    return st >= 0;
}

// ------------------------------------------------------------------------------------------- GP_Regression
GP_Regression::GP_Regression(const int &input_dim, c_kernel *k, c_meanfunc *m, c_likelihood *l, c_inference_hip *i, c_prior *p)
    : dim(input_dim), kernel(k), meanfunc(m), likfunc(l), inffunc(i), prior(p) {}
GP_Regression::~GP_Regression() { delete[] chol_alpha; delete[] chol_factor_inv; }   // nullptr-safe (the reference's is not)

void GP_Regression::train(const bool &flag_grad, const vector<int> &meta, const vector<float> &x, const vector<float> &y) {
    // The reference allocates two N*N float buffers per call (gp_regression.cpp:111-117) that only predict() reads.
    // The device keeps alpha / L^-1 resident, so nothing is copied here; predict() goes to the device.
    flag_trained = false;
    float *na = nullptr, *nl = nullptr;
    flag_trained = inffunc->compute_nlml(flag_grad, meta, x, y, kernel, meanfunc, likfunc, prior, na, nl, beta,
                                         nlm_likelihood, dnlm_likelihood);
    if (!flag_trained) std::cout << "Warning: current inference failed in train()!!" << std::endl;
}

vector<vector<float>> GP_Regression::predict(const vector<int> &meta, const vector<int> &meta2, const vector<float> &x,
                                             const vector<float> &y, const vector<float> &x2) {
    vector<float> mean, var;
    float *na = nullptr, *nl = nullptr;
    double f; vector<double> g; float b;
    // (re)select the patient on the device; the factorisation itself happens inside medgp_fit_predict
    if (!flag_trained) train(false, meta, x, y);
    else inffunc->compute_nlml(false, meta, x, y, kernel, meanfunc, likfunc, prior, na, nl, b, f, g);
    inffunc->predict(kernel, likfunc, meta2, x2, mean, var);
    return {mean, var};
}

// ------------------------------------------------------------------------------------------- c_objective_one
bool c_objective_one::compute_objective(const bool &flag_grad, const vector<double> &input_parameter, double &objective_value,
                                        vector<double> &gradients, c_kernel *&input_kernel, c_meanfunc *&input_meanfunc,
                                        c_likelihood *&input_likfunc, c_inference_hip *&input_inffunc, c_prior *&input_prior) {
    if ((int)obj_x.size() > 2) {   // ref: c_objective_one.cpp:51
        c_hyperparam hyp(input_parameter, input_kernel->get_kernel_hyp_num(), input_meanfunc->get_meanfunc_hyp_num(),
                         input_likfunc->get_likfunc_hyp_num());
        input_kernel->set_kernel_hyp(hyp.get_hyp_cov());
        input_meanfunc->set_meanfunc_hyp(hyp.get_hyp_mean());
        input_likfunc->set_likfunc_hyp(hyp.get_hyp_lik());
        GP_Regression curr_gpr_model(1, input_kernel, input_meanfunc, input_likfunc, input_inffunc, input_prior);
        curr_gpr_model.train(flag_grad, obj_meta, obj_x, obj_y);
        if (!curr_gpr_model.get_flag_trained()) return false;
        objective_value = curr_gpr_model.get_neg_log_mlikelihood();
        if (flag_grad) gradients = curr_gpr_model.get_dneg_log_mlikelihood();
        return true;
    }
    return false;
}

}  // namespace medgp
