// micro-benchmark: which instruction classes of a wave are held up by a co-resident wave that streams fp64 MFMAs on the same
// SIMD?  Workgroup 0 (one wave) runs 4096 instructions of ONE class; workgroup 256 lands on the same CU (see kernels_cholinv_la.h)
// and streams v_mfma_f64_16x16x4_f64 on all four SIMDs; workgroups 1..255 sleep.  Reported: cycles per instruction alone / beside.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
#define REP64(x) x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x
__global__ void __launch_bounds__(256, 2) k(unsigned long long *cyc, int *flag, int cls, int neighbour, double *sink) {
    __shared__ double lds[1024];
    const int tid = threadIdx.x, lane = tid & 63;
    lds[tid] = tid; lds[tid + 256] = tid; lds[tid + 512] = 1; lds[tid + 768] = 2;
    __syncthreads();
    if (blockIdx.x == 0) {
        if (tid >= 64) return;
        double a = 1.0 + lane, b = 0.999, c = 0.5, d = 0.25, e = 0.125;
        int ia = lane, ib = 3, ic = 5, id = 7;
        v4d m0 = {0, 0, 0, 0}, m1 = m0;
        const double *lp = lds + lane;
        unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < 16; it++) {
            switch (cls) {
            case 0: REP64(asm volatile("v_fma_f64 %0, %0, %4, %0\n\tv_fma_f64 %1, %1, %4, %1\n\tv_fma_f64 %2, %2, %4, %2\n\tv_fma_f64 %3, %3, %4, %3" : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b));) break;
            case 1: REP64(asm volatile("v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4" : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id) : "v"(lane));) break;
            case 2: REP64(asm volatile("v_fma_f32 %0, %0, %4, %0\n\tv_fma_f32 %1, %1, %4, %1\n\tv_fma_f32 %2, %2, %4, %2\n\tv_fma_f32 %3, %3, %4, %3" : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id) : "v"(lane));) break;
            case 3: REP64(asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:512\n\tds_read_b64 %2, %4 offset:1024\n\tds_read_b64 %3, %4 offset:1536\n\ts_waitcnt lgkmcnt(0)" : "=v"(a), "=v"(c), "=v"(d), "=v"(e) : "v"((int)(size_t)lp & 0xffff) : "memory");) break;
            case 4: REP64(asm volatile("s_add_u32 s40, s40, 1\n\ts_add_u32 s40, s40, 1\n\ts_add_u32 s40, s40, 1\n\ts_add_u32 s40, s40, 1" ::: "s40", "scc");) break;
            case 5: REP64(m0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, m0, 0, 0, 0); m1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, m1, 0, 0, 0); m0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, m0, 0, 0, 0); m1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, m1, 0, 0, 0);) break;
            case 6: REP64(asm volatile("v_readlane_b32 s41, %0, 3\n\tv_readlane_b32 s41, %0, 5\n\tv_readlane_b32 s41, %0, 7\n\tv_readlane_b32 s41, %0, 9" :: "v"(ia) : "s41");) break;
            case 7: REP64(asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %4\n\tv_mov_b32 %2, %4\n\tv_mov_b32 %3, %4" : "=v"(ia), "=v"(ic), "=v"(id), "=v"(ib) : "v"(lane));) break;
            }
        }
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
        sink[lane] = a + c + d + e + ia + ib + ic + id + m0[0] + m1[1];
        if (lane == 0) { cyc[0] = t1 - t0; __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        return;
    }
    if (blockIdx.x == 256 && neighbour) {
        v4d c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        double a = 1.0 + lane * 1e-3, b = 0.5;
        while (!__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
            }
        }
        sink[256 + tid] = c0[0] + c1[1] + c2[2] + c3[3];
        return;
    }
    if (tid < 64) while (!__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) __builtin_amdgcn_s_sleep(64);
}
int main() {
    unsigned long long *dc; int *flag; double *sink;
    hipMalloc(&dc, 64); hipMalloc(&flag, 4); hipMalloc(&sink, 8 * 1024);
    const char *names[8] = {"v_fma_f64", "v_add_u32", "v_fma_f32", "ds_read_b64 x4 + wait", "s_add_u32", "v_mfma_f64_16x16x4", "v_readlane_b32", "v_mov_b32"};
    for (int cls = 0; cls < 8; cls++) {
        double res[2];
        for (int nb = 0; nb < 2; nb++) {
            unsigned long long c = 0;
            for (int it = 0; it < 2; it++) {
                hipMemset(flag, 0, 4);
                hipLaunchKernelGGL(k, dim3(257), dim3(256), 0, 0, dc, flag, cls, nb, sink);
                hipDeviceSynchronize();
            }
            hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
            res[nb] = (double)c / (16.0 * 64 * 4);
        }
        printf("%-24s: %7.1f cycles/instruction alone, %7.1f beside a dense fp64 MFMA stream\n", names[cls], res[0], res[1]);
    }
    return 0;
}
