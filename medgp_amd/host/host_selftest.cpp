// host_selftest.cpp -- exercises the C++ host adapter the way main_one_train.cpp:103-118 / 228-238 and
// main_one_test.cpp:386-399 use the reference classes.  Reads a problem file, evaluates through
// c_objective_one::compute_objective and GP_Regression::predict, writes the results for the test to compare
// with the oracle.
//   in : int32 D,N,Q,R,H,prior_mode,nstar | int32 meta[N] | float x[N] | float y[N] | double theta[H]
//        | int32 meta2[nstar] | float x2[nstar]
//   out: int32 ok | double nlml | double grad[H] | float mean[nstar] | float var[nstar] | float alpha[N] | float beta
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "medgp_host.hpp"
using namespace medgp;

int main(int argc, char **argv) {
    if (argc != 3) { fprintf(stderr, "usage: host_selftest in.bin out.bin\n"); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    int32_t hdr[7];
    if (fread(hdr, 4, 7, f) != 7) return 2;
    const int D = hdr[0], N = hdr[1], Q = hdr[2], R = hdr[3], H = hdr[4], prior_mode = hdr[5], nstar = hdr[6];
    std::vector<int> meta(N), meta2(nstar);
    std::vector<float> x(N), y(N), x2(nstar);
    std::vector<double> theta(H);
    if (fread(meta.data(), 4, N, f) != (size_t)N || fread(x.data(), 4, N, f) != (size_t)N || fread(y.data(), 4, N, f) != (size_t)N ||
        fread(theta.data(), 8, H, f) != (size_t)H || fread(meta2.data(), 4, nstar, f) != (size_t)nstar ||
        fread(x2.data(), 4, nstar, f) != (size_t)nstar) return 2;
    fclose(f);

    // object set of run_model_LMC_SM (ref: main_one_train.cpp:103-118)
    std::vector<int> kparam = {Q, D, R};
    c_kernel kernel(7, kparam);
    c_inference_hip inffunc(1);
    c_meanfunc_zero meanfunc;
    c_likelihood likfunc(D);
    c_prior prior(kernel.get_kernel_hyp_num(), 0, D);
    if (prior_mode == 2) prior.setup_param(7, kparam, 2, {0.01f, 0.01f});
    c_kernel *kptr = &kernel; c_meanfunc *mptr = &meanfunc; c_likelihood *lptr = &likfunc;
    c_inference_hip *iptr = &inffunc; c_prior *pptr = &prior;

    c_objective_one obj(7, kparam, meta, x, y);
    double nlml = 0.0;
    std::vector<double> grad;
    bool ok = obj.compute_objective(true, theta, nlml, grad, kptr, mptr, lptr, iptr, pptr);
    // nlml-only evaluation must give the same value (HOT LOOP A of main_one_train.cpp:228-253)
    double nlml0 = 0.0;
    std::vector<double> g0;
    bool ok0 = obj.compute_objective(false, theta, nlml0, g0, kptr, mptr, lptr, iptr, pptr);
    if (ok && (!ok0 || nlml0 != nlml)) { fprintf(stderr, "nlml-only mismatch %.17g vs %.17g\n", nlml0, nlml); return 3; }

    // train(false) + predict as main_one_test.cpp:386-399
    std::vector<float> mean(nstar), var(nstar), alpha(N, 0.f);
    float beta = 0.f;
    if (ok) {
        GP_Regression gpr(1, kptr, mptr, lptr, iptr, pptr);
        gpr.train(false, meta, x, y);
        auto post = gpr.predict(meta, meta2, x, y, x2);
        mean = post[0]; var = post[1];
        float *ap = alpha.data(), *lp = nullptr;
        double fdum; std::vector<double> gdum;
        ok = ok && inffunc.compute_nlml(true, meta, x, y, kptr, mptr, lptr, pptr, ap, lp, beta, fdum, gdum);
    }
    grad.resize(H, 0.0);
    FILE *o = fopen(argv[2], "wb");
    int32_t iok = ok ? 1 : 0;
    fwrite(&iok, 4, 1, o); fwrite(&nlml, 8, 1, o); fwrite(grad.data(), 8, H, o);
    fwrite(mean.data(), 4, nstar, o); fwrite(var.data(), 4, nstar, o); fwrite(alpha.data(), 4, N, o); fwrite(&beta, 4, 1, o);
    fclose(o);
    printf("host_selftest: ok=%d nlml=%.12f status=%d\n", iok, nlml, inffunc.last_status());
    return ok ? 0 : 1;
}
