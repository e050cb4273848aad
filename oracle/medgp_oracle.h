/*
 * medgp_oracle.h -- CPU restatement (fp64, plain C) of the MedGP per-patient
 * nlml + gradient hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under medgp_amd/ (the product) may include,
 * link, dlopen or call this.  Allowed users: tests/, __graft_entry__.smoke(),
 * and bench.py's cpu_baseline leg -- as the checker / reported baseline, never
 * as the thing measured or shipped.
 *
 * PARITY PINNING STATUS: "partially pinned" -- see the header of
 * medgp_oracle.c and DESIGN.md section 3.
 *
 * All citations "ref:" are file:line into /root/reference/medgpc/src/.
 */
#ifndef MEDGP_ORACLE_H
#define MEDGP_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* kernel_index values follow ref: main_one_train.cpp:85-93 */
#define MEDGP_ORACLE_KERNEL_SE      0
#define MEDGP_ORACLE_KERNEL_LMC_SM  7
#define MEDGP_ORACLE_KERNEL_SM      8

/* ref: util/global_settings.h:6 -- the reference's truncated pi literal */
#define MEDGP_ORACLE_REF_PI 3.14159265

/* gradient algorithm of the oracle */
#define MEDGP_ORACLE_GRAD_PER_HYPER 0 /* the reference's loop: one N x N dK/dtheta_h per hyper (c_kernel_LMC_SM.cpp:222-325) */
#define MEDGP_ORACLE_GRAD_BLOCKED   1 /* same sums regrouped into Q block reductions S_q */

/* number of likelihood hypers / covariance hypers / total for a kernel */
int medgp_oracle_num_lik(int kernel_index, int D);
int medgp_oracle_num_cov(int kernel_index, int Q, int D, int R);
int medgp_oracle_num_hyp(int kernel_index, int Q, int D, int R);

/* B_q = A_q A_q^T + diag(kappa_q), q-major, each D x D row-major.
 * theta_cov = raw covariance hypers (log-domain for mu, v, kappa).
 * ref: kernel/c_kernel_LMC_SM.cpp:51-62 (exp transform), :72-115 (B_q). */
void medgp_oracle_lmc_coregional(int Q, int D, int R, const double *theta_cov, double *B);

/* one basis-kernel value k(r^2; mu, v) with mu, v already transformed.
 * ref: kernel/c_kernel_LMC_SM.cpp:374-378 */
double medgp_oracle_sm_k(double rsq, double mu, double v, double pi);

/* Gram matrix with the noise diagonal added once: K (n x n, row-major, symmetric).
 * theta = full raw hyper vector [lik | cov].  meta may be NULL for SE/SM.
 * ref: kernel/c_kernel_LMC_SM.cpp:152-196, c_kernel_SE.cpp:72-89, c_kernel_SM.cpp:75-110,
 *      inference/c_inference_exact.cpp:88-92 */
int medgp_oracle_gram(int kernel_index, int Q, int D, int R, double pi,
                      int n, const int32_t *meta, const float *t,
                      const double *theta, double *K);

/* prior log-densities and the prior stage on their own (ref: prior/c_prior.cpp:383-421, inference/c_inference_prior.cpp:60-150):
 * type 1 normal (p0 mean, p1 variance), 2 laplace (p0 location, p1 scale), anything else: lp = dlp = 0.
 * apply_prior: hval[H] = transformed hyper values in theta order, nlml / grad updated in place (grad may be NULL when !flag_grad). */
void medgp_oracle_prior_lik(int type, double x, float p0, float p1, double pi, double *lp, double *dlp);
void medgp_oracle_apply_prior(int H, const double *hval, double pi, int flag_grad, const uint8_t *pflag, const int32_t *ptype,
                              const uint8_t *pexp, const float *p0, const float *p1, double *nlml, double *grad);

/* The operator: c_inference_prior::compute_nlml restated in fp64.
 *   prior_* arrays have one entry per hyper in theta order (may all be NULL = no prior):
 *     prior_flag  : 1 = prior active for this hyper            (ref: prior/c_prior.h:35-38)
 *     prior_type  : -1 none, 0 clamp, 1 normal, 2 laplace      (ref: prior/c_prior.h:50-53)
 *     prior_exp   : 1 = multiply d(log p) by the transformed hyper (ref: c_inference_prior.cpp:113-118)
 *     prior_p0/p1 : float parameters (mean, variance) / (location, scale) (ref: prior/c_prior.h:46-48)
 *   outputs (each may be NULL): nlml, grad[H], alpha[n] (= K^-1 y), Linv[n*n] (row-major lower,
 *     strict upper zero; ref: c_inference_exact.cpp:130-143), beta (= y^T K^-1 y).
 *   status: 0..10 = number of jitter rounds applied (ref: c_inference_exact.cpp:96-111), -1 = failed.
 * returns 1 on success (the reference's `true`), 0 on failure (Cholesky failed after 10 jitters or n <= 2,
 *   ref: util/c_objective_one.cpp:51,79-81).
 */
int medgp_oracle_nlml_grad(int kernel_index, int Q, int D, int R, double pi,
                           int n, const int32_t *meta, const float *t, const float *y,
                           const double *theta, int flag_grad, int grad_mode, int nthreads,
                           const uint8_t *prior_flag, const int32_t *prior_type,
                           const uint8_t *prior_exp, const float *prior_p0, const float *prior_p1,
                           double *nlml, double *grad, double *alpha, double *Linv, double *beta,
                           int32_t *status);

/* GP_Regression::train(false) + predict restated in fp64.
 * ref: core/gp_regression.cpp:128-214, kernel/c_kernel_LMC_SM.cpp:329-372 (cross), :122-150 (diag) */
int medgp_oracle_fit_predict(int kernel_index, int Q, int D, int R, double pi,
                             int n, const int32_t *meta, const float *t, const float *y,
                             const double *theta,
                             int nstar, const int32_t *meta2, const float *t2,
                             double *mean, double *var, int32_t *status);

#ifdef __cplusplus
}
#endif
#endif
