// ref_prior_inference_dump.cpp -- TEST INFRASTRUCTURE (build container only; never shipped, never linked into the product).
//
// Runs the reference's OWN c_inference_prior::compute_nlml (ref: inference/c_inference_prior.cpp:25-154, compiled unmodified, with
// the reference's own c_prior from prior/c_prior.cpp) and prints its inputs and outputs as JSON.  That function is where the prior
// terms of row a19 are applied: nlml -= lp, the exp chain rule  dnlml -= hyp * dlp,  the clamp  dnlml = 0.
//
// WHAT IS AND IS NOT THE REFERENCE HERE -- read before trusting the fixture:
//   * c_inference_prior.cpp and c_prior.cpp are the reference's translation units, compiled as they lie under /root/reference.
//   * c_inference_prior::compute_nlml first calls c_inference_exact::compute_nlml (the dense GP evaluation) and reads the transformed
//     hypers through c_kernel::get_kernel_hyp / c_likelihood::get_likfunc_hyp / c_meanfunc::get_meanfunc_hyp.  Those four members live
//     in files that include <mkl.h> (inference/c_inference_exact.cpp, kernel/c_kernel.cpp, likelihoods/c_likelihood.cpp,
//     mean/c_meanfunc.cpp) and cannot be compiled here.  They are declared by the reference's headers (which do compile) and DEFINED
//     BELOW AS TEST DOUBLES: the "exact inference" returns the nlml / gradient / success flag this driver hands it, the getters return
//     the vectors this driver stored.  No MKL header or library is imitated; nothing below pretends to be the reference's GP algebra.
//   * Therefore this fixture pins exactly one thing: given (base nlml, base gradient, transformed hyper values, a c_prior object built
//     by the reference's own setup_param / init_test_prior / the variational-EM writes), the numbers the reference's prior stage
//     produces.  The base evaluation (rows a2-a18) is NOT pinned by it.
//   * The transformed hyper values are formed here as the reference's setters form them (exp of every likelihood hyper,
//     ref: likelihoods/c_likelihood.cpp:38-43; A raw and exp of everything behind it, ref: kernel/c_kernel_LMC_SM.cpp:51-59) -- three
//     lines restated, stated so here.
#include <cmath>
#include <cstdio>
#include <iostream>
#include <vector>

#include "core/gp_model_include.h"

// ---- test doubles for members whose reference definitions sit behind <mkl.h> ------------------------------------------------------
c_kernel::c_kernel() {}
vector<double> c_kernel::get_kernel_hyp() { return kernel_hyp; }
c_likelihood::c_likelihood() {}
vector<double> c_likelihood::get_likfunc_hyp() { return likfunc_hyp; }
c_meanfunc::c_meanfunc() {}
vector<double> c_meanfunc::get_meanfunc_hyp() { return meanfunc_hyp; }
namespace {
struct Base { bool ok; double nlml; vector<double> dnlml; } g_base;
struct test_kernel : c_kernel { void put(const vector<double> &v) { kernel_hyp = v; } };
struct test_lik : c_likelihood { void put(const vector<double> &v) { likfunc_hyp = v; } };
struct test_mean : c_meanfunc { void put(const vector<double> &v) { meanfunc_hyp = v; } };
}  // namespace
c_inference_exact::c_inference_exact(const int &thread_num) { inf_thread_num = thread_num; }
bool c_inference_exact::compute_nlml(const bool &flag_grad, const vector<int> &, const vector<float> &, const vector<float> &, c_kernel *,
                                     c_meanfunc *, c_likelihood *, c_prior *, float *&, float *&, float &, double &nlml, vector<double> &dnlml) {
    if (!g_base.ok) return false;
    nlml = g_base.nlml;
    if (flag_grad) dnlml = g_base.dnlml;
    return true;
}
// ---------------------------------------------------------------------------------------------------------------------------------

static FILE *out = nullptr;
template <typename T>
static void arr(const char *name, const std::vector<T> &v, const char *fmt, bool last = false) {
    fprintf(out, "\"%s\": [", name);
    for (size_t i = 0; i < v.size(); i++) { fprintf(out, fmt, v[i]); if (i + 1 < v.size()) fputc(',', out); }
    fprintf(out, "]%s", last ? "" : ", ");
}

// deterministic theta in the optimiser's variables: [log sigma_d | A raw | log mu | log v | log kappa]; some A entries exactly 0
static std::vector<double> make_theta(int Q, int D, int R, int salt) {
    const int H = D + Q * (D * R + 2 + D);
    std::vector<double> th(H);
    for (int i = 0; i < H; i++) {
        const double u = std::fmod(0.6180339887498949 * (double)(i + 1) + 0.137 * (double)salt, 1.0);   // low-discrepancy in [0, 1)
        if (i < D) th[i] = std::log(0.15 + 0.25 * u);
        else if (i < D + Q * D * R) th[i] = ((i + salt) % 11 == 5) ? 0.0 : (3.0 * u - 1.5) * 0.4;
        else if (i < D + Q * D * R + Q) th[i] = std::log(1.0 / (12.0 + 60.0 * u));
        else if (i < D + Q * D * R + 2 * Q) th[i] = std::log(1.0 / (2.0 * 3.14159265 * (6.0 + 66.0 * u)));
        else th[i] = std::log((0.1 + 0.4 * u) * 0.02);
    }
    return th;
}

struct Case { const char *name; int Q, D, R, mode; bool test_clamp, varem_writes, flag_grad, ok; };

int main() {
    out = stdout;
    std::cout.rdbuf(std::cerr.rdbuf());
    const float eta = 0.01f, beta_lam = 0.01f;
    const Case cases[] = {
        {"PT_INR_mode2", 5, 2, 2, 2, false, false, true, true},
        {"PT_INR_mode2_varem", 5, 2, 2, 2, false, true, true, true},
        {"PT_INR_mode2_testclamp", 5, 2, 2, 2, true, false, true, true},
        {"PT_INR_mode0_testclamp", 5, 2, 2, 0, true, false, true, true},
        {"PT_INR_mode0", 5, 2, 2, 0, false, false, true, true},
        {"PT_INR_mode2_nograd", 5, 2, 2, 2, false, true, false, true},
        {"PT_INR_mode2_failed", 5, 2, 2, 2, false, false, true, false},
        {"all24_mode2_varem", 5, 24, 8, 2, false, true, true, true},
        {"all24_mode2_testclamp", 5, 24, 8, 2, true, true, true, true},
        {"D64_mode2_varem", 5, 64, 8, 2, false, true, true, true},
    };
    const int ncase = (int)(sizeof(cases) / sizeof(cases[0]));
    fprintf(out, "{\"eta\": %.9g, \"beta_lam\": %.9g, \"cases\": [", (double)eta, (double)beta_lam);
    for (int ci = 0; ci < ncase; ci++) {
        const Case &c = cases[ci];
        const int Q = c.Q, D = c.D, R = c.R, ncov = Q * (D * R + 2 + D), nlik = D, H = nlik + ncov;
        std::vector<int> kp = {Q, D, R};
        std::vector<float> pp = {eta, beta_lam};
        std::vector<double> th = make_theta(Q, D, R, ci);
        c_prior prior(ncov, 0, nlik);
        prior.setup_param(7, kp, c.mode, pp);
        if (c.varem_writes) {
            // what one outer iteration of the variational-EM loop leaves in the object (ref: util/c_optimizer_varEM.cpp:140-161):
            // fix_param_cov[index] = {0, (float)psi}; psi == 0 -> type_cov[index] = 0 and the optimiser's a = 0
            for (int i = 0; i < Q * D * R; i++) {
                const double psi = (i % 13 == 7) ? 0.0 : 0.05 + 0.9 * std::fmod(0.7548776662466927 * (double)(i + 1), 1.0);
                prior.set_cov_varEM_one(psi, i);
                if (prior.get_cov_varEM_one(i) == 0.0) { prior.type_cov[i] = 0; th[nlik + i] = 0.0; }
                prior.fix_param_cov[i][0] = 0;
                prior.fix_param_cov[i][1] = prior.get_cov_varEM_one(i);
            }
        }
        if (c.test_clamp) {   // main_one_test's prior: clamp the A entries that are exactly zero in the mode kernel
            std::vector<double> mode(th);
            prior.init_test_prior(7, kp, mode);
        }
        // transformed hyper values, as the reference's setters store them (see the header)
        std::vector<double> lik(nlik), cov(ncov);
        for (int i = 0; i < nlik; i++) lik[i] = std::exp(th[i]);
        for (int i = 0; i < ncov; i++) cov[i] = (i < Q * D * R) ? th[nlik + i] : std::exp(th[nlik + i]);
        test_kernel k; k.put(cov);
        test_lik l; l.put(lik);
        test_mean m; m.put(std::vector<double>());
        // the base evaluation handed to the reference's prior stage
        g_base.ok = c.ok;
        g_base.nlml = 123.456 + (double)ci;
        g_base.dnlml.assign(H, 0.0);
        for (int i = 0; i < H; i++) g_base.dnlml[i] = 1.0 + 0.25 * std::sin(0.37 * (double)(i + 1) + (double)ci);
        c_inference_prior inf(1);
        float *ca = nullptr, *cf = nullptr, beta = 0.f;
        double nlml = -777.0;
        std::vector<double> dnlml(H, -777.0);
        std::vector<int> meta;
        std::vector<float> x, y;
        c_prior *pptr = &prior;
        const bool ok = inf.compute_nlml(c.flag_grad, meta, x, y, &k, &m, &l, pptr, ca, cf, beta, nlml, dnlml);
        // prior descriptor in theta order (lik | cov), flat
        std::vector<int> flag, type, ex, plen;
        std::vector<double> p0, p1;
        for (int i = 0; i < H; i++) {
            const bool isl = i < nlik;
            const int j = isl ? i : i - nlik;
            flag.push_back((isl ? prior.flag_lik[j] : prior.flag_cov[j]) ? 1 : 0);
            type.push_back(isl ? prior.type_lik[j] : prior.type_cov[j]);
            ex.push_back((isl ? prior.exp_lik[j] : prior.exp_cov[j]) ? 1 : 0);
            const std::vector<float> &fp = isl ? prior.fix_param_lik[j] : prior.fix_param_cov[j];
            plen.push_back((int)fp.size());
            p0.push_back(fp.size() > 0 ? (double)fp[0] : 0.0);
            p1.push_back(fp.size() > 1 ? (double)fp[1] : 1.0);
        }
        fprintf(out, "{\"name\": \"%s\", \"Q\": %d, \"D\": %d, \"R\": %d, \"mode\": %d, \"flag_grad\": %d, \"base_ok\": %d, \"ok\": %d, ", c.name, Q, D, R, c.mode,
                c.flag_grad ? 1 : 0, c.ok ? 1 : 0, ok ? 1 : 0);
        arr("theta", th, "%.17g");
        std::vector<double> hval(lik);
        hval.insert(hval.end(), cov.begin(), cov.end());
        arr("hval", hval, "%.17g");
        arr("prior_flag", flag, "%d"); arr("prior_type", type, "%d"); arr("prior_exp", ex, "%d"); arr("prior_len", plen, "%d");
        arr("prior_p0", p0, "%.17g"); arr("prior_p1", p1, "%.17g");
        fprintf(out, "\"base_nlml\": %.17g, ", g_base.nlml);
        arr("base_dnlml", g_base.dnlml, "%.17g");
        fprintf(out, "\"nlml\": %.17g, ", nlml);
        arr("dnlml", dnlml, "%.17g", true);
        fprintf(out, "}%s", ci + 1 < ncase ? ", " : "");
    }
    fprintf(out, "]}\n");
    return 0;
}
