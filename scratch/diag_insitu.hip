// micro-benchmark: the one-wave 64x64 factor (diag_factor_wave) ALONE on its CU and next to a workgroup that streams fp64 MFMAs
// on all four SIMDs of the same CU.  Placement: workgroup 0 factors; workgroups 1..255 sleep (they keep the other CUs occupied);
// workgroup 256 is the first one the dispatcher places on workgroup 0's CU (see the parked-neighbour note in kernels_cholinv_la.h).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include "../medgp_amd/csrc/kernels_core.h"
#include "../medgp_amd/csrc/kernels_cholinv.h"
struct Sm { double D[64][66], X[64][66]; alignas(16) double dv[64 + 128]; double logdet; int fail; };
__global__ void __launch_bounds__(256, 2) k(const double *A, unsigned long long *cyc, int *flag, int reps, int neighbour_mode, double *sink) {
    __shared__ Sm sm;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if ((blockIdx.x == 0 || blockIdx.x == 256) && tid == 0) {
        // s_getreg_b32 HW_ID (id 4) and XCC_ID (id 20): where did this workgroup land, and when
        unsigned hw = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 4), xcc = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 20);
        const int o = blockIdx.x == 0 ? 8 : 12;
        cyc[o] = hw; cyc[o + 1] = xcc; cyc[o + 2] = wall_clock64();
    }
    if (blockIdx.x == 0) {
        unsigned long long tot = 0;
        for (int r = 0; r < reps; r++) {
            for (int e = tid; e < 64 * 64; e += 256) { sm.D[e >> 6][e & 63] = A[e]; sm.X[e >> 6][e & 63] = 0.0; }
            if (tid == 0) { sm.fail = 0; sm.logdet = 0.0; }
            __syncthreads();
            if (neighbour_mode == 6 && wave == 0) __builtin_amdgcn_s_setprio(3);
            unsigned long long t0 = __builtin_amdgcn_s_memtime();
            if (wave == 0) diag_factor_wave((ld_t *)&sm.D[0][0], (ld_t *)&sm.X[0][0], (ld_t *)sm.dv, (li_t *)&sm.fail, (ld_t *)&sm.logdet, lane);
            unsigned long long t1 = __builtin_amdgcn_s_memtime();
            if (wave == 0) tot += t1 - t0;
            __syncthreads();
        }
        if (tid == 0) { cyc[0] = tot / reps; __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        return;
    }
    if (blockIdx.x == 256 && neighbour_mode) {
        // neighbour: independent fp64 MFMA chains (mode 1: all four waves; mode 2: only wave 0, i.e. the factoring wave's SIMD mate idle)
        if (neighbour_mode == 6) __builtin_amdgcn_s_setprio(0);
        if (neighbour_mode == 2 && wave == 0) { while (!__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) __builtin_amdgcn_s_sleep(32); return; }
        v4d c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        double a = 1.0 + lane * 1e-3, b = 0.5;
        unsigned long long n = 0;
        while (!__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            if (neighbour_mode == 7) {
#pragma unroll
                for (int i = 0; i < 64; i++) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
                n += 64;
                continue;
            }
#pragma unroll
            for (int i = 0; i < 16; i++) {
                c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
            }
            n += 64;
            if (neighbour_mode == 3) __builtin_amdgcn_s_sleep(1);            // yield 64 cycles every 64 MFMAs (1.5 % of the stream)
            if (neighbour_mode == 4) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7");
            if (neighbour_mode == 5) __builtin_amdgcn_s_setprio(0);
        }
        sink[tid] = c0[0] + c1[1] + c2[2] + c3[3];
        if (lane == 0) cyc[2 + wave] = n;
        return;
    }
    // everybody else keeps a slot of its CU busy doing nothing
    if (tid < 64) while (!__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) __builtin_amdgcn_s_sleep(64);
}
int main() {
    std::vector<double> A(64 * 64);
    for (int i = 0; i < 64; i++) for (int j = 0; j < 64; j++) A[i * 64 + j] = std::exp(-0.01 * (i - j) * (i - j)) + (i == j ? 0.5 : 0.0);
    double *dA, *sink; unsigned long long *dc; int *flag;
    hipMalloc(&dA, 8 * 4096); hipMalloc(&dc, 128); hipMalloc(&flag, 4); hipMalloc(&sink, 8 * 256);
    hipMemcpy(dA, A.data(), 8 * 4096, hipMemcpyHostToDevice);
    const char *names[8] = {"alone on its CU", "neighbour streams MFMA on 4 SIMDs", "neighbour: one wave sleeps", "neighbour: s_sleep 1 every 64 MFMAs",
                            "neighbour: s_nop 7 every 64 MFMAs", "neighbour: s_setprio 0 every 64 MFMAs", "neighbour streams, factor wave at s_setprio 3", "neighbour: dependent MFMA chain (1 accumulator)"};
    for (int mode = 0; mode < 8; mode++) {
        unsigned long long c[16] = {0};
        for (int it = 0; it < 2; it++) {
            hipMemset(flag, 0, 4); hipMemset(dc, 0, 128);
            hipLaunchKernelGGL(k, dim3(257), dim3(256), 0, 0, dA, dc, flag, 20, mode, sink);
            hipDeviceSynchronize();
        }
        hipMemcpy(c, dc, 128, hipMemcpyDeviceToHost);
        printf("   wg0: HW_ID %08llx (se %llu sh %llu cu %llu) XCC %llx | wg256: HW_ID %08llx (se %llu sh %llu cu %llu) XCC %llx, started %.1f us after wg0\n",
               c[8], (c[8] >> 13) & 7, (c[8] >> 12) & 1, (c[8] >> 8) & 15, c[9] & 0xf, c[12], (c[12] >> 13) & 7, (c[12] >> 12) & 1, (c[12] >> 8) & 15, c[13] & 0xf,
               ((double)c[14] - (double)c[10]) / 100.0);
        printf("%-42s: %llu cycles per factor; neighbour MFMAs per wave during the 20 factors: %llu %llu %llu %llu\n", names[mode], c[0], c[2], c[3], c[4], c[5]);
    }
    return 0;
}
