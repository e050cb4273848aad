// Checks that the BLGP immediate of v_mfma_f64_16x16x4_f64 acts as NEG[a, b, c]: build with hipcc --offload-arch=gfx950, run on the device.
// Observed (MI355X): "negA mismatches 0": -(a b) + c is bit-identical to the product with an explicitly negated operand.
#include <hip/hip_runtime.h>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void k(const double *a, const double *b, double *out) {
    int l = threadIdx.x;
    v4d c = {0.0, 0.0, 0.0, 0.0};
    v4d c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[l], b[l], c, 0, 0, 0);
    v4d c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[l], b[l], c, 0, 0, 1);
    v4d c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[l], b[l], c0, 0, 0, 4);
    for (int r = 0; r < 4; r++) { out[l * 12 + r] = c0[r]; out[l * 12 + 4 + r] = c1[r]; out[l * 12 + 8 + r] = c2[r]; }
}
int main() {
    double *a, *b, *o; hipMalloc(&a, 64 * 8); hipMalloc(&b, 64 * 8); hipMalloc(&o, 64 * 12 * 8);
    double ha[64], hb[64], ho[64 * 12];
    for (int i = 0; i < 64; i++) { ha[i] = 1.0 + 0.01 * i; hb[i] = 2.0 - 0.02 * i; }
    hipMemcpy(a, ha, 512, hipMemcpyHostToDevice); hipMemcpy(b, hb, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, o); hipMemcpy(ho, o, 64 * 12 * 8, hipMemcpyDeviceToHost);
    int bad1 = 0, bad2 = 0;
    for (int l = 0; l < 64; l++) for (int r = 0; r < 4; r++) { if (ho[l * 12 + 4 + r] != -ho[l * 12 + r]) bad1++; if (ho[l * 12 + 8 + r] != 0.0) bad2++; }
    printf("negA mismatches %d  (ab - c0 with negC) nonzero %d  sample %g %g %g\n", bad1, bad2, ho[0], ho[4], ho[8]);
    return 0;
}
