// ref_prior_dump.cpp -- TEST INFRASTRUCTURE (build container only; never shipped, never linked into the product).
//
// Driver for the two translation units of the reference that compile here UNMODIFIED with plain g++ (no MKL, no rapidjson, no
// stand-in headers):   /root/reference/medgpc/src/prior/c_prior.cpp   and   core/c_hyperparam.cpp.
// It calls the reference's own c_prior / c_hyperparam objects and prints what they hold as JSON on stdout;
// tests/golden/make_golden.py::ref_prior turns that into tests/golden/ref_prior.json, which pins
//   * row a19 (prior terms): c_prior::setup_param / setup_hier_gamma_prior (ref: prior/c_prior.cpp:196-279), init_test_prior (the
//     clamp of exact zeros of the mode kernel, :118-140), prior_lik_normal / prior_lik_laplace (:383-421);
//   * the start state of the variational-EM loop of row f1: init_cov_varEM / init_cov_varEM_fix (:109-116, :234-241);
//   * row a2 (theta split): c_hyperparam::set_hyp_all / get_hyp_all (ref: core/c_hyperparam.cpp:68-122).
// Built by `make -C oracle ref` into oracle/_ref/ (git-ignored).  The reference prints progress lines on std::cout; this
// driver redirects std::cout to std::cerr while it calls the reference and writes its own JSON to the real stdout.
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <vector>

#include "core/c_hyperparam.h"
#include "prior/c_prior.h"

static FILE *out = nullptr;

template <typename T>
static void arr(const char *name, const std::vector<T> &v, const char *fmt, bool last = false) {
    fprintf(out, "\"%s\": [", name);
    for (size_t i = 0; i < v.size(); i++) { fprintf(out, fmt, v[i]); if (i + 1 < v.size()) fputc(',', out); }
    fprintf(out, "]%s", last ? "" : ", ");
}
static void arrb(const char *name, const std::vector<bool> &v) {
    fprintf(out, "\"%s\": [", name);
    for (size_t i = 0; i < v.size(); i++) { fprintf(out, "%d", v[i] ? 1 : 0); if (i + 1 < v.size()) fputc(',', out); }
    fprintf(out, "], ");
}
// fix_param_*: vector<vector<float>> of length 0 or 2 -> two arrays (NaN-free: -1e30 marks "no parameter")
static void fixp(const char *name, const std::vector<std::vector<float>> &v) {
    std::vector<double> p0, p1;
    std::vector<int> len;
    for (const auto &e : v) {
        len.push_back((int)e.size());
        p0.push_back(e.size() > 0 ? (double)e[0] : -1e30);
        p1.push_back(e.size() > 1 ? (double)e[1] : -1e30);
    }
    char nm[64];
    snprintf(nm, sizeof nm, "%s_len", name); arr(nm, len, "%d");
    snprintf(nm, sizeof nm, "%s_p0", name); arr(nm, p0, "%.17g");
    snprintf(nm, sizeof nm, "%s_p1", name); arr(nm, p1, "%.17g");
}

static void dump_prior(c_prior &p, const char *key, bool last = false) {
    fprintf(out, "\"%s\": {", key);
    arrb("flag_lik", p.flag_lik); arrb("flag_cov", p.flag_cov); arrb("flag_mean", p.flag_mean);
    arrb("exp_lik", p.exp_lik); arrb("exp_cov", p.exp_cov); arrb("exp_mean", p.exp_mean);
    arr("type_lik", p.type_lik, "%d"); arr("type_cov", p.type_cov, "%d"); arr("type_mean", p.type_mean, "%d");
    fixp("fix_lik", p.fix_param_lik); fixp("fix_cov", p.fix_param_cov); fixp("fix_mean", p.fix_param_mean);
    arr("cov_varEM", p.get_cov_varEM_all(), "%.17g");
    arr("cov_varEM_fix", p.get_cov_varEM_fix_all(), "%.17g", true);
    fprintf(out, "}%s", last ? "" : ", ");
}

int main() {
    out = stdout;
    std::cout.rdbuf(std::cerr.rdbuf());   // the reference's progress lines go to stderr
    const int shapes[3][3] = {{5, 2, 2}, {5, 24, 8}, {5, 64, 8}};   // (Q, D, R): BASELINE configs 1-2, 3-4, 5
    const float eta = 0.01f, beta_lam = 0.01f;                       // scripts/opt_prior2.json ("eta", "beta_lam")
    fprintf(out, "{\"eta\": %.9g, \"beta_lam\": %.9g, \"shapes\": [", (double)eta, (double)beta_lam);
    for (int s = 0; s < 3; s++) {
        const int Q = shapes[s][0], D = shapes[s][1], R = shapes[s][2];
        const int ncov = Q * (D * R + 2 + D), nmean = 0, nlik = D;   // ref: kernel/c_kernel_LMC_SM.cpp:64-70, gaussianMO: D
        std::vector<int> kp = {Q, D, R};
        std::vector<float> pp = {eta, beta_lam};                     // c_experiment::get_prior_hyp order (ref: dataio/c_experiment.cpp prior_hyp)
        fprintf(out, "{\"Q\": %d, \"D\": %d, \"R\": %d, \"ncov\": %d, \"nmean\": %d, \"nlik\": %d, ", Q, D, R, ncov, nmean, nlik);
        {   // mode 0: no regularisation -- as main_one_train.cpp:108-115 constructs it, then setup_param(7, kp, 0, pp)
            c_prior p(ncov, nmean, nlik);
            p.setup_param(7, kp, 0, pp);
            dump_prior(p, "mode0");
        }
        {   // mode 2: hierarchical gamma prior
            c_prior p(ncov, nmean, nlik);
            p.setup_param(7, kp, 2, pp);
            dump_prior(p, "mode2");
            // without prior parameters (the defaults 50.0 / 0.5 of :234-241, :264-270)
            c_prior pd(ncov, nmean, nlik);
            pd.setup_param(7, kp, 2, std::vector<float>());
            dump_prior(pd, "mode2_default");
            // a kernel index other than 7: "prior will not be effective" (:203-206)
            c_prior pk(3, 0, 1);
            pk.setup_param(0, kp, 2, pp);
            dump_prior(pk, "mode2_kernel0");
            // the test-side clamp: a mode kernel with exact zeros among the A entries (main_one_test.cpp: init_test_prior)
            std::vector<double> mode(nlik + ncov + nmean);
            for (size_t i = 0; i < mode.size(); i++) mode[i] = 0.125 * (double)((int)(i % 7) - 3);   // every 7th entry (i % 7 == 3) is exactly 0.0
            c_prior pt(ncov, nmean, nlik);
            pt.setup_param(7, kp, 2, pp);
            pt.init_test_prior(7, kp, mode);
            arr("test_mode", mode, "%.17g");
            dump_prior(pt, "mode2_test");
            c_prior pt0(ncov, nmean, nlik);
            pt0.setup_param(7, kp, 0, pp);
            pt0.init_test_prior(7, kp, mode);
            dump_prior(pt0, "mode0_test");
            // get_one_prior_flag / get_one_prior_type in theta order (lik | cov | mean)
            std::vector<int> gf, gt;
            for (int i = 0; i < nlik + ncov + nmean; i++) { gf.push_back(pt.get_one_prior_flag(i) ? 1 : 0); gt.push_back(pt.get_one_prior_type(i)); }
            arr("test_flag_theta_order", gf, "%d");
            arr("test_type_theta_order", gt, "%d");
            // get_one_lik_cov through the object: one A entry (normal), one kappa entry (laplace), one mu entry (no prior), a clamped one
            std::vector<double> xs = {-1.5, -0.25, 0.0, 0.01, 0.3, 2.0};
            std::vector<int> idx = {0, 3, Q * D * R, Q * (D * R + 2), ncov - 1};
            std::vector<double> lp, dlp;
            for (int i : idx)
                for (double x : xs) { std::vector<double> l = pt.get_one_lik_cov(x, i); lp.push_back(l[0]); dlp.push_back(l[1]); }
            arr("lik_cov_idx", idx, "%d"); arr("lik_cov_x", xs, "%.17g"); arr("lik_cov_lp", lp, "%.17g"); arr("lik_cov_dlp", dlp, "%.17g");
        }
        {   // c_hyperparam: split of one theta vector [lik | cov | mean] and the round trip
            std::vector<double> th(nlik + ncov + 2);
            for (size_t i = 0; i < th.size(); i++) th[i] = 0.001 * (double)(i + 1) - 0.5;
            c_hyperparam h(th, ncov, 2, nlik);
            arr("hyp_in", th, "%.17g"); arr("hyp_lik", h.get_hyp_lik(), "%.17g"); arr("hyp_cov", h.get_hyp_cov(), "%.17g");
            arr("hyp_mean", h.get_hyp_mean(), "%.17g");
            std::vector<int> nn = {h.get_num_hyp_lik(), h.get_num_hyp_cov(), h.get_num_hyp_mean(), h.get_num_hyp_all()};
            arr("hyp_counts", nn, "%d");
            arr("hyp_all", h.get_hyp_all(), "%.17g", true);
        }
        fprintf(out, "}%s", s < 2 ? ", " : "");
    }
    fprintf(out, "], ");
    {   // prior_lik_normal / prior_lik_laplace on a grid incl. x == m (the Laplace kink, :404-406), float parameters as stored
        c_prior p(1, 0, 1);
        const float params[5][2] = {{0.0f, 1.0f}, {0.0f, 0.01f}, {0.25f, 0.5f}, {-1.0f, 2.5f}, {0.0f, 0.37f}};
        std::vector<double> xs = {-3.0, -1.0, -0.25, -1e-9, 0.0, 1e-9, 0.01, 0.25, 0.37, 1.0, 2.5, 7.0};
        std::vector<double> m, s, nlp, ndlp, llp, ldlp;
        for (int k = 0; k < 5; k++) {
            std::vector<float> pr = {params[k][0], params[k][1]};
            for (double x : xs) {
                std::vector<double> a = p.prior_lik_normal(x, pr), b = p.prior_lik_laplace(x, pr);
                m.push_back((double)pr[0]); s.push_back((double)pr[1]);
                nlp.push_back(a[0]); ndlp.push_back(a[1]); llp.push_back(b[0]); ldlp.push_back(b[1]);
            }
        }
        fprintf(out, "\"lik_grid\": {");
        arr("x", xs, "%.17g"); arr("p0", m, "%.17g"); arr("p1", s, "%.17g");
        arr("normal_lp", nlp, "%.17g"); arr("normal_dlp", ndlp, "%.17g"); arr("laplace_lp", llp, "%.17g"); arr("laplace_dlp", ldlp, "%.17g", true);
        fprintf(out, "}}\n");
    }
    return 0;
}
