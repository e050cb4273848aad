// micro-benchmark: one 256-thread workgroup per CU streams 64-row x 64-column (512 B per row) fp64 blocks of a row-major matrix with
// row stride S bytes -- the access pattern of the look-ahead GEMM tasks -- for several S, against the same bytes laid out tile-major.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v2d __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(256) k(const double *A, size_t stride_d, int nblk_cols, int iters, double *out, int tile_major) {
    const int tid = threadIdx.x, wg = blockIdx.x;
    // workgroup wg reads block row (wg % 32), walking over block columns
    const int srow = tid >> 5, piece = tid & 31;
    v2d acc = {0.0, 0.0};
    for (int it = 0; it < iters; it++)
        for (int bc = 0; bc < nblk_cols; bc++) {
            v2d v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int r = srow + 8 * u;
                const double *p = tile_major ? A + ((size_t)(wg % 32) * nblk_cols + bc) * 4096 + r * 64 + 2 * piece
                                             : A + (size_t)(64 * (wg % 32) + r) * stride_d + 64 * bc + 2 * piece;
                v[u] = *(const v2d *)p;
            }
#pragma unroll
            for (int u = 0; u < 8; u++) acc += v[u];
        }
    if (acc[0] == 1.2345) out[0] = acc[1];
}
int main() {
    const int nb = 32;   // 32 x 32 blocks of 64 x 64 = N 2048
    for (int mode = 0; mode < 5; mode++) {
        size_t stride_d = (mode == 0) ? 2048 : (mode == 1) ? 2048 + 64 : (mode == 2) ? 4096 : (mode == 3) ? 4096 + 64 : 2048;
        const int tile_major = (mode == 4);
        size_t elems = (size_t)64 * nb * stride_d + 4096;
        double *A, *out;
        hipMalloc(&A, elems * 8); hipMalloc(&out, 64);
        hipMemset(A, 0, elems * 8);
        for (int grid : {1, 32, 256}) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, A, stride_d, nb, 2, out, tile_major);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, A, stride_d, nb, 20, out, tile_major);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double bytes = (double)grid * 20 * nb * 32768.0;
            printf("%s stride %6zu B  grid %3d: %8.1f GB/s per workgroup, %8.1f GB/s total\n", tile_major ? "tile-major" : "row-major ", stride_d * 8, grid,
                   bytes / grid / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 1e9);
        }
        hipFree(A); hipFree(out);
    }
    return 0;
}
