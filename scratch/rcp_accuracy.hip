// accuracy of v_rcp_f64 / v_rsq_f64 seeds on gfx950 and of one / two Newton steps (decides the length of the pivot chain of diag16)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double *x, double *o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double p = x[i];
    double r0 = __builtin_amdgcn_rcp(p);
    double r1 = fma(fma(-p, r0, 1.0), r0, r0);
    double r2 = fma(fma(-p, r1, 1.0), r1, r1);
    const double e0 = fma(-p, r0, 1.0);
    const double r3 = fma(r0, fma(e0, e0, e0), r0);   // one cubic step: r0 (1 + e + e^2), three dependent fma
    double y0 = __builtin_amdgcn_rsq(p);
    o[5 * i + 0] = r0; o[5 * i + 1] = r1; o[5 * i + 2] = r2; o[5 * i + 3] = y0; o[5 * i + 4] = r3;
}
int main() {
    const int n = 1 << 22;
    std::vector<double> h(n), out(5 * (size_t)n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; double u = (s >> 11) * (1.0 / 9007199254740992.0); h[i] = ldexp(1.0 + u, (int)(s % 41) - 20); }
    double *dx, *dout;
    hipMalloc(&dx, n * 8); hipMalloc(&dout, 5 * (size_t)n * 8);
    hipMemcpy(dx, h.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
    hipMemcpy(out.data(), dout, 5 * (size_t)n * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0, es = 0, e3 = 0; long exact1 = 0, exact2 = 0, exact3 = 0;
    for (int i = 0; i < n; i++) {
        long double t = 1.0L / (long double)h[i];
        double tr = (double)t;
        e0 = fmax(e0, fabs((double)((out[5 * i] - t) / t)));
        e1 = fmax(e1, fabs((double)((out[5 * i + 1] - t) / t)));
        e2 = fmax(e2, fabs((double)((out[5 * i + 2] - t) / t)));
        exact1 += out[5 * i + 1] == tr; exact2 += out[5 * i + 2] == tr;
        e3 = fmax(e3, fabs((double)((out[5 * i + 4] - t) / t))); exact3 += out[5 * i + 4] == tr;
        long double ts = 1.0L / sqrtl((long double)h[i]);
        es = fmax(es, fabs((double)((out[5 * i + 3] - ts) / ts)));
    }
    printf("v_rcp_f64 max rel err %.3e (2^%.1f); after 1 Newton step %.3e (2^%.1f), correctly rounded %.4f %%; after 2 steps %.3e, correctly rounded %.4f %%\n",
           e0, log2(e0), e1, log2(e1), 100.0 * exact1 / n, e2, 100.0 * exact2 / n);
    printf("one cubic step (3 dependent fma): max rel err %.3e, correctly rounded %.4f %%\n", e3, 100.0 * exact3 / n);
    printf("v_rsq_f64 max rel err %.3e (2^%.1f)\n", es, log2(es));
    return 0;
}
