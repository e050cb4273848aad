// micro-benchmark: 64x64 diagonal-block factor + inverse, one wave (diag_factor_wave) vs four waves (diag_factor_wg)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include "../medgp_amd/csrc/kernels_core.h"
#include "../medgp_amd/csrc/kernels_cholinv.h"
struct Sm { double D[64][66], X[64][66]; alignas(16) double dv[64 + 128]; double logdet; int fail; };
template <int MODE>
__global__ void __launch_bounds__(256) k(const double *A, double *outL, double *outX, unsigned long long *cyc, int reps) {
    __shared__ Sm sm;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned long long tot = 0;
    for (int r = 0; r < reps; r++) {
        for (int e = tid; e < 64 * 64; e += 256) { sm.D[e >> 6][e & 63] = A[e]; sm.X[e >> 6][e & 63] = 0.0; }
        if (tid == 0) { sm.fail = 0; sm.logdet = 0.0; }
        __syncthreads();
        unsigned long long t0 = __builtin_amdgcn_s_memtime();
        if (MODE == 0) { if (wave == 0) diag_factor_wave((ld_t *)&sm.D[0][0], (ld_t *)&sm.X[0][0], (ld_t *)sm.dv, (li_t *)&sm.fail, (ld_t *)&sm.logdet, lane); __syncthreads(); }
        else diag_factor_wg((ld_t *)&sm.D[0][0], (ld_t *)&sm.X[0][0], (ld_t *)sm.dv, (li_t *)&sm.fail, (ld_t *)&sm.logdet, wave, lane);
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
        tot += t1 - t0;
        __syncthreads();
    }
    if (tid == 0) { cyc[0] = tot / reps; cyc[1] = sm.fail; }
    for (int e = tid; e < 64 * 64; e += 256) { outL[e] = sm.D[e >> 6][e & 63]; outX[e] = sm.X[e >> 6][e & 63]; }
    if (tid == 0) outL[64 * 64] = sm.logdet;
}
int main() {
    std::vector<double> A(64 * 64);
    for (int i = 0; i < 64; i++) for (int j = 0; j < 64; j++) A[i * 64 + j] = std::exp(-0.01 * (i - j) * (i - j)) + (i == j ? 0.5 : 0.0);
    double *dA, *dL, *dX; unsigned long long *dc;
    hipMalloc(&dA, 8 * 4096); hipMalloc(&dL, 8 * 4097); hipMalloc(&dX, 8 * 4096); hipMalloc(&dc, 16);
    hipMemcpy(dA, A.data(), 8 * 4096, hipMemcpyHostToDevice);
    std::vector<double> L0(4097), X0(4096), L1(4097), X1(4096);
    unsigned long long c[2];
    for (int mode = 0; mode < 2; mode++) {
        for (int it = 0; it < 2; it++) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(256), 0, 0, dA, dL, dX, dc, 20);
            else hipLaunchKernelGGL(k<1>, dim3(1), dim3(256), 0, 0, dA, dL, dX, dc, 20);
            hipDeviceSynchronize();
        }
        hipMemcpy(c, dc, 16, hipMemcpyDeviceToHost);
        hipMemcpy(mode ? L1.data() : L0.data(), dL, 8 * 4097, hipMemcpyDeviceToHost);
        hipMemcpy(mode ? X1.data() : X0.data(), dX, 8 * 4096, hipMemcpyDeviceToHost);
        printf("mode %d: %llu cycles per factor (fail %llu)\n", mode, c[0], c[1]);
    }
    double dl = 0, dx = 0;
    for (int i = 0; i < 64; i++) for (int j = 0; j <= i; j++) { dl = fmax(dl, fabs(L0[i * 64 + j] - L1[i * 64 + j])); }
    for (int e = 0; e < 4096; e++) dx = fmax(dx, fabs(X0[e] - X1[e]));
    // check L X = I
    double res = 0;
    for (int i = 0; i < 64; i++) for (int j = 0; j < 64; j++) { double s = 0; for (int k = 0; k < 64; k++) s += (k <= i ? L1[i * 64 + k] : 0.0) * X1[k * 64 + j]; res = fmax(res, fabs(s - (i == j))); }
    printf("max |L0-L1| %.3e  max |X0-X1| %.3e  |L X - I| %.3e  logdet %.12f %.12f\n", dl, dx, res, L0[4096], L1[4096]);
    return 0;
}
