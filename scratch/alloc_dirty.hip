// micro-benchmark: does hipMalloc get slow once the device memory it is given has been USED before (by this or an earlier process)?
//   alloc_dirty dirty <GB>      allocate <GB>, write all of it, free, exit
//   alloc_dirty time <GB> [n]   n times: hipMalloc(<GB>) timed, written, freed
// Build: hipcc --offload-arch=gfx950 -O2 scratch/alloc_dirty.hip -o scratch/alloc_dirty
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAILED %s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_touch(double *p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = 1.0;
}

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    const size_t GB = (size_t)1 << 30, bytes = (size_t)atoll(argv[2]) * GB;
    CK(hipSetDevice(0));
    double t00 = now();
    CK(hipFree(nullptr));
    printf("[%s %s] runtime init %.1f ms\n", argv[1], argv[2], (now() - t00) * 1e3);
    const int n = argc > 3 ? atoi(argv[3]) : 1;
    for (int r = 0; r < (strcmp(argv[1], "dirty") ? n : 1); r++) {
        double *p = nullptr;
        double t0 = now();
        CK(hipMalloc(&p, bytes));
        double t1 = now();
        hipLaunchKernelGGL(k_touch, dim3(4096), dim3(256), 0, nullptr, p, bytes / 8);
        CK(hipDeviceSynchronize());
        double t2 = now();
        CK(hipFree(p));
        double t3 = now();
        printf("  hipMalloc(%zu GB) %9.3f ms (%.4f s/GB)   touch %8.3f ms   hipFree %8.3f ms\n", bytes / GB, (t1 - t0) * 1e3, (t1 - t0) / ((double)bytes / GB),
               (t2 - t1) * 1e3, (t3 - t2) * 1e3);
    }
    return 0;
}
