// micro-benchmark: what does device memory cost on this platform, per GB, by the way it is obtained?
//   hipMalloc (first time, after a hipFree of the same size), first touch / second touch by a kernel, hipFree,
//   VMM (hipMemAddressReserve + hipMemCreate + hipMemMap + hipMemSetAccess in 1 GB / 256 MB granules), hipMallocAsync from a pool
//   with a release threshold.  Build: hipcc --offload-arch=gfx950 -O2 scratch/alloc_cost.hip -o scratch/alloc_cost ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAILED %s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_touch(double *p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = 1.0;
}

static int touch(double *p, size_t bytes, const char *tag) {
    double t0 = now();
    hipLaunchKernelGGL(k_touch, dim3(4096), dim3(256), 0, nullptr, p, bytes / 8);
    CK(hipDeviceSynchronize());
    double t1 = now();
    printf("    %-28s %8.3f ms  (%.1f GB/s)\n", tag, (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
    return 0;
}

int main() {
    CK(hipSetDevice(0));
    CK(hipFree(nullptr));
    { double *w; CK(hipMalloc(&w, 1 << 20)); hipLaunchKernelGGL(k_touch, dim3(16), dim3(256), 0, nullptr, w, (size_t)1 << 17); CK(hipDeviceSynchronize()); CK(hipFree(w)); }
    const size_t GB = (size_t)1 << 30;
    for (size_t bytes : {GB / 16, GB / 4, GB, 4 * GB, 16 * GB}) {
        printf("hipMalloc %.3f GB\n", (double)bytes / GB);
        double *p = nullptr;
        double t0 = now();
        CK(hipMalloc(&p, bytes));
        double t1 = now();
        printf("    %-28s %8.3f ms  (%.3f s/GB)\n", "hipMalloc", (t1 - t0) * 1e3, (t1 - t0) / ((double)bytes / GB));
        if (touch(p, bytes, "first touch")) return 1;
        if (touch(p, bytes, "second touch")) return 1;
        t0 = now();
        CK(hipFree(p));
        t1 = now();
        printf("    %-28s %8.3f ms\n", "hipFree", (t1 - t0) * 1e3);
        t0 = now();
        CK(hipMalloc(&p, bytes));
        t1 = now();
        printf("    %-28s %8.3f ms  (%.3f s/GB)\n", "hipMalloc again", (t1 - t0) * 1e3, (t1 - t0) / ((double)bytes / GB));
        if (touch(p, bytes, "first touch again")) return 1;
        CK(hipFree(p));
    }
    // many small allocations: 64 x 64 MB
    {
        std::vector<double *> v(64);
        double t0 = now();
        for (auto &p : v) CK(hipMalloc(&p, GB / 16));
        double t1 = now();
        printf("64 x hipMalloc(64 MB): %.3f ms (%.3f s/GB)\n", (t1 - t0) * 1e3, (t1 - t0) / 4.0);
        for (auto &p : v) CK(hipFree(p));
    }
    // VMM
    {
        hipMemAllocationProp prop{};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        size_t gran = 0;
        hipError_t e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
        printf("VMM granularity: %s, %zu bytes\n", hipGetErrorString(e), gran);
        if (e == hipSuccess && gran) {
            void *base = nullptr;
            double t0 = now();
            e = hipMemAddressReserve(&base, 64 * GB, 0, nullptr, 0);
            double t1 = now();
            printf("    hipMemAddressReserve(64 GB): %s  %.3f ms\n", hipGetErrorString(e), (t1 - t0) * 1e3);
            if (e == hipSuccess) {
                for (size_t chunk : {GB / 4, GB}) {
                    std::vector<hipMemGenericAllocationHandle_t> hs;
                    size_t off = 0;
                    double tc = 0, tm = 0, ta = 0;
                    const int nchunk = (int)(4 * GB / chunk);
                    bool ok = true;
                    for (int i = 0; i < nchunk && ok; i++) {
                        hipMemGenericAllocationHandle_t h;
                        double a = now();
                        e = hipMemCreate(&h, chunk, &prop, 0);
                        double b = now();
                        if (e != hipSuccess) { printf("    hipMemCreate: %s\n", hipGetErrorString(e)); ok = false; break; }
                        e = hipMemMap((char *)base + off, chunk, 0, h, 0);
                        double c = now();
                        if (e != hipSuccess) { printf("    hipMemMap: %s\n", hipGetErrorString(e)); ok = false; break; }
                        hipMemAccessDesc ad{};
                        ad.location = prop.location;
                        ad.flags = hipMemAccessFlagsProtReadWrite;
                        e = hipMemSetAccess((char *)base + off, chunk, &ad, 1);
                        double d = now();
                        if (e != hipSuccess) { printf("    hipMemSetAccess: %s\n", hipGetErrorString(e)); ok = false; break; }
                        tc += b - a; tm += c - b; ta += d - c;
                        hs.push_back(h);
                        off += chunk;
                    }
                    if (ok) {
                        printf("  VMM 4 GB in %d chunks of %.2f GB: create %.3f ms, map %.3f ms, setaccess %.3f ms  (%.3f s/GB)\n", nchunk, (double)chunk / GB,
                               tc * 1e3, tm * 1e3, ta * 1e3, (tc + tm + ta) / 4.0);
                        if (touch((double *)base, off, "first touch (VMM)")) return 1;
                        if (touch((double *)base, off, "second touch (VMM)")) return 1;
                    }
                    double t2 = now();
                    size_t o2 = 0;
                    for (auto h : hs) { (void)hipMemUnmap((char *)base + o2, chunk); (void)hipMemRelease(h); o2 += chunk; }
                    printf("    unmap + release: %.3f ms\n", (now() - t2) * 1e3);
                }
                (void)hipMemAddressFree(base, 64 * GB);
            }
        }
    }
    // stream-ordered pool
    {
        hipMemPool_t pool;
        hipError_t e = hipDeviceGetDefaultMemPool(&pool, 0);
        printf("default mem pool: %s\n", hipGetErrorString(e));
        if (e == hipSuccess) {
            uint64_t thr = UINT64_MAX;
            (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr);
            for (int rep = 0; rep < 2; rep++) {
                double *p = nullptr;
                double t0 = now();
                e = hipMallocAsync((void **)&p, 4 * GB, nullptr);
                CK(hipDeviceSynchronize());
                double t1 = now();
                printf("    hipMallocAsync(4 GB) #%d: %s %.3f ms\n", rep, hipGetErrorString(e), (t1 - t0) * 1e3);
                if (e != hipSuccess) break;
                if (touch(p, 4 * GB, "first touch (pool)")) return 1;
                t0 = now();
                CK(hipFreeAsync(p, nullptr));
                CK(hipDeviceSynchronize());
                printf("    hipFreeAsync: %.3f ms\n", (now() - t0) * 1e3);
            }
        }
    }
    // one big block last: 48 GB
    {
        double *p = nullptr;
        double t0 = now();
        hipError_t e = hipMalloc(&p, 48 * GB);
        double t1 = now();
        printf("hipMalloc(48 GB): %s %.3f ms (%.3f s/GB)\n", hipGetErrorString(e), (t1 - t0) * 1e3, (t1 - t0) / 48.0);
        if (e == hipSuccess) { if (touch(p, 48 * GB, "first touch 48 GB")) return 1; if (touch(p, 48 * GB, "second touch 48 GB")) return 1; t0 = now(); CK(hipFree(p)); printf("    hipFree %.3f ms\n", (now() - t0) * 1e3); }
    }
    return 0;
}
