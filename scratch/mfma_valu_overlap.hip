// microbenchmark: do fp64 MFMA and fp64 VALU FMA streams of two waves on one SIMD overlap?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(512) k(int mode, int iters, double *out) {
    const int w = threadIdx.x >> 6;
    double acc = threadIdx.x * 1e-9;
    if (w < 4) {
        if (mode & 1) {
            v4d c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
            double a = 1.0 + threadIdx.x * 1e-6, b = 1.0 - threadIdx.x * 1e-6;
            for (int i = 0; i < iters; i++) {
                c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
            }
            acc += c0[0] + c1[1] + c2[2] + c3[3];
        }
    } else {
        if (mode & 2) {
            double x0 = acc, x1 = acc + 1, x2 = acc + 2, x3 = acc + 3, x4 = acc + 4, x5 = acc + 5, x6 = acc + 6, x7 = acc + 7;
            const double m = 0.999999, p = 1e-7;
            for (int i = 0; i < iters; i++) {   // 16 independent-ish fp64 FMAs per iteration = same issue time as 4 MFMAs? (4 cycles each)
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    x0 = fma(x0, m, p); x1 = fma(x1, m, p); x2 = fma(x2, m, p); x3 = fma(x3, m, p);
                    x4 = fma(x4, m, p); x5 = fma(x5, m, p); x6 = fma(x6, m, p); x7 = fma(x7, m, p);
                }
            }
            acc += x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
        }
    }
    if (acc == 123.456) out[0] = acc;
}
int main() {
    double *d; hipMalloc(&d, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int mode = 1; mode <= 3; mode++) {
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, 100, d);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, iters, d);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // per SIMD: MFMA wave issues 4*iters MFMAs (64 cycles each); VALU wave issues 64*iters FMAs (4 cycles each)
        printf("mode %d (%s): %.3f ms  | MFMA-only ideal %.3f ms, VALU-only ideal %.3f ms at 2.4 GHz\n", mode,
               mode == 1 ? "MFMA" : mode == 2 ? "VALU" : "both", ms, 4.0 * iters * 64 / 2.4e6, 64.0 * iters * 4 / 2.4e6);
    }
    return 0;
}
