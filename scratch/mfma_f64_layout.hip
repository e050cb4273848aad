// micro-test: v_mfma_f64_16x16x4_f64 operand / result lane maps and issue rate on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void k_layout(const double* A, const double* B, double* C) {  // A[16][4], B[4][16], C[16][16]
    int l = threadIdx.x;
    double a = A[(l & 15) * 4 + (l >> 4)];
    double b = B[(l >> 4) * 16 + (l & 15)];
    v4d c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++) C[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];
}
__global__ void k_rate(double* out, int iters) {
    int l = threadIdx.x;
    double a = 1.0 + l * 1e-3, b = 0.5 - l * 1e-3;
    v4d c0 = {0,0,0,0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters; i++) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + l] = c0[0] + c1[1] + c2[2] + c3[3];
}
__global__ void k_rate_dep(double* out, int iters) {
    int l = threadIdx.x;
    double a = 1.0 + l * 1e-3, b = 0.5 - l * 1e-3;
    v4d c0 = {0,0,0,0};
    for (int i = 0; i < iters; i++) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + l] = c0[0];
}
__global__ void k_exp(double* out, int iters) {
    double x = -1e-3 * (threadIdx.x + 1), acc = 0;
    for (int i = 0; i < iters; i++) { acc += exp(x); x -= 1e-4; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
__global__ void k_fma(double* out, int iters) {
    double x = 1e-3 * (threadIdx.x + 1), a0 = 0, a1 = 1, a2 = 2, a3 = 3;
    for (int i = 0; i < iters; i++) { a0 = a0 * x + 1.0; a1 = a1 * x + 1.0; a2 = a2 * x + 1.0; a3 = a3 * x + 1.0; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}
int main() {
    std::vector<double> A(64), B(64), C(256), R(256, 0.0);
    for (int i = 0; i < 16; i++) for (int k = 0; k < 4; k++) A[i * 4 + k] = 1 + i * 0.37 + k * 1.7;
    for (int k = 0; k < 4; k++) for (int j = 0; j < 16; j++) B[k * 16 + j] = 2 - j * 0.11 + k * k * 0.5;
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) for (int k = 0; k < 4; k++) R[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
    double *dA, *dB, *dC, *dO;
    hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dC, 2048); hipMalloc(&dO, 8 * 1024 * 1024);
    hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
    k_layout<<<1, 64>>>(dA, dB, dC);
    hipMemcpy(C.data(), dC, 2048, hipMemcpyDeviceToHost);
    double err = 0; for (int i = 0; i < 256; i++) err = fmax(err, fabs(C[i] - R[i]));
    printf("layout max err %g (%s)\n", err, err < 1e-12 ? "OK" : "WRONG");
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char* name, void (*k)(double*, int), int blocks, int threads, int iters, double flop_per_thread_iter) {
        k<<<blocks, threads>>>(dO, 10); hipDeviceSynchronize();
        hipEventRecord(e0); k<<<blocks, threads>>>(dO, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double tot = (double)blocks * threads * iters * flop_per_thread_iter;
        printf("%s: %d blocks x %d thr: %.3f ms, %.2f TFLOP/s-equiv\n", name, blocks, threads, ms, tot / ms / 1e9);
    };
    // mfma: 4 per iter, each 16*16*4*2 = 2048 flop per wave => per thread 2048*4/64
    timeit("mfma_f64 indep x4, 1 wave/SIMD", k_rate, 256, 256, 20000, 2048.0 * 4 / 64);
    timeit("mfma_f64 indep x4, 2 wave/SIMD", k_rate, 256, 512, 20000, 2048.0 * 4 / 64);
    timeit("mfma_f64 dep chain, 1 wave/SIMD", k_rate_dep, 256, 256, 20000, 2048.0 * 4 / 64);
    timeit("mfma_f64 dep chain, 2 wave/SIMD", k_rate_dep, 256, 512, 20000, 2048.0 * 4 / 64);
    timeit("dfma x4 chains, 4 wave/SIMD", k_fma, 256, 1024, 20000, 8);
    timeit("exp(double) ocml, 4 wave/SIMD (1 'flop' = 1 exp)", k_exp, 256, 1024, 2000, 1);
    return 0;
}
