/* asan_driver.c -- sanitizer run of the oracle (`make -C oracle asan`): both gradient forms, priors, predict, the n <= 2 guard
 * and a failing factorisation, on small deterministic inputs.  TEST INFRASTRUCTURE ONLY (see medgp_oracle.h). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "medgp_oracle.h"

static double lcg(unsigned long long *s) { *s = *s * 6364136223846793005ULL + 1442695040888963407ULL; return (double)(*s >> 11) * (1.0 / 9007199254740992.0); }

int main(void) {
    const int Q = 3, D = 3, R = 2, n = 45, H = medgp_oracle_num_hyp(MEDGP_ORACLE_KERNEL_LMC_SM, Q, D, R);
    unsigned long long s = 12345;
    int32_t *meta = malloc(sizeof(int32_t) * n);
    float *t = malloc(sizeof(float) * n), *y = malloc(sizeof(float) * n);
    double *th = malloc(sizeof(double) * H), *g0 = malloc(sizeof(double) * H), *g1 = malloc(sizeof(double) * H);
    double *alpha = malloc(sizeof(double) * n), *linv = malloc(sizeof(double) * n * n);
    uint8_t *pf = calloc(H, 1), *pe = calloc(H, 1);
    int32_t *pt = malloc(sizeof(int32_t) * H);
    float *p0 = calloc(H, sizeof(float)), *p1 = malloc(sizeof(float) * H);
    for (int i = 0; i < n; i++) { meta[i] = i * D / n; t[i] = (float)(200.0 * lcg(&s)); y[i] = (float)(2.0 * lcg(&s) - 1.0); }
    for (int h = 0; h < H; h++) {
        th[h] = (h < D) ? log(0.15 + 0.25 * lcg(&s)) : (h < D + Q * D * R) ? (3.0 * lcg(&s) - 1.5) * 0.4 : log(0.02 + 0.05 * lcg(&s));
        pt[h] = (h >= D && h < D + Q * D * R) ? 1 : -1; pf[h] = pt[h] == 1; p1[h] = 1.0f;
    }
    pt[D + 1] = 0;   /* one clamped entry */
    double nl0, nl1, beta;
    int32_t st;
    int ok = medgp_oracle_nlml_grad(7, Q, D, R, MEDGP_ORACLE_REF_PI, n, meta, t, y, th, 1, MEDGP_ORACLE_GRAD_PER_HYPER, 2, pf, pt, pe, p0, p1, &nl0, g0, alpha, linv, &beta, &st);
    ok &= medgp_oracle_nlml_grad(7, Q, D, R, MEDGP_ORACLE_REF_PI, n, meta, t, y, th, 1, MEDGP_ORACLE_GRAD_BLOCKED, 1, pf, pt, pe, p0, p1, &nl1, g1, NULL, NULL, NULL, &st);
    double gmax = 0, dmax = 0;
    for (int h = 0; h < H; h++) { if (fabs(g0[h]) > gmax) gmax = fabs(g0[h]); if (fabs(g0[h] - g1[h]) > dmax) dmax = fabs(g0[h] - g1[h]); }
    if (!ok || st != 0 || !(fabs(nl0 - nl1) <= 1e-12 * fabs(nl0)) || !(dmax <= 1e-9 * gmax)) { printf("FAIL gradient forms %d %d %g %g %g\n", ok, st, nl0, nl1, dmax); return 1; }
    /* n <= 2 guard (ref: util/c_objective_one.cpp:51) and a singular matrix (zero noise on duplicates): both report failure */
    ok = medgp_oracle_nlml_grad(7, Q, D, R, MEDGP_ORACLE_REF_PI, 2, meta, t, y, th, 1, MEDGP_ORACLE_GRAD_BLOCKED, 1, NULL, NULL, NULL, NULL, NULL, &nl0, g0, NULL, NULL, NULL, &st);
    if (ok || st != -1) { printf("FAIL guard\n"); return 1; }
    for (int i = 1; i < n; i++) t[i] = t[0];
    for (int i = 0; i < n; i++) meta[i] = 0;
    for (int d = 0; d < D; d++) th[d] = -800.0;
    ok = medgp_oracle_nlml_grad(7, Q, D, R, MEDGP_ORACLE_REF_PI, n, meta, t, y, th, 0, MEDGP_ORACLE_GRAD_BLOCKED, 1, NULL, NULL, NULL, NULL, NULL, &nl0, NULL, NULL, NULL, NULL, &st);
    if (ok || st != -1) { printf("FAIL singular %d %d\n", ok, st); return 1; }
    free(meta); free(t); free(y); free(th); free(g0); free(g1); free(alpha); free(linv); free(pf); free(pe); free(pt); free(p0); free(p1);
    printf("oracle sanitizer run ok\n");
    return 0;
}
