// stress: growth of a VMM arena by chunks of varying size, access set over [base, mapped) after every new chunk, relocation to a
// range twice as large when outgrown; every byte ever written is verified at the end.  (hipcc --offload-arch=gfx950 ...)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <time.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAILED line %d %s: %s\n", __LINE__, #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_fill(unsigned *p, size_t n, unsigned seed) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; for (; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = seed + (unsigned)i; }
__global__ void k_check(const unsigned *p, size_t n, unsigned seed, int *bad) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; for (; i < n; i += (size_t)gridDim.x * blockDim.x) if (p[i] != seed + (unsigned)i) { if (atomicAdd(bad, 1) == 0) { bad[1] = (int)p[i]; bad[2] = (int)(seed + (unsigned)i); bad[3] = (int)(i >> 18); } } }
int main(int argc, char **argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int mode = argc > 1 ? atoi(argv[1]) : 0;   // 0: access over the whole mapped range; 1: access per new chunk
    CK(hipSetDevice(0)); CK(hipFree(nullptr));
    hipMemAllocationProp prop{}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc ad{}; ad.location = prop.location; ad.flags = hipMemAccessFlagsProtReadWrite;
    const size_t MB = 1 << 20;
    int *bad; CK(hipMalloc(&bad, 16)); CK(hipMemset(bad, 0, 16));
    const int keep = argc > 2 ? atoi(argv[2]) : 0;   // 1: never release physical memory between arenas; 2: release, then wait 3 s
    srand(12345);
    for (int arena = 0; arena < 6; arena++) {
        char *base = nullptr; size_t reserved = 2 * MB, mapped = 0;
        std::vector<std::pair<hipMemGenericAllocationHandle_t, size_t>> ch;
        CK(hipMemAddressReserve((void **)&base, reserved, 2 * MB, nullptr, 0));
        int nreloc = 0;
        for (int step = 0; step < 40; step++) {
            const size_t grow = (size_t)(2 * (1 + rand() % 40)) * MB;
            if (mapped + grow > reserved) {
                size_t want = reserved * 2; while (want < mapped + grow) want *= 2;
                char *nb = nullptr;
                CK(hipMemAddressReserve((void **)&nb, want, 2 * MB, nullptr, 0));
                CK(hipDeviceSynchronize());
                size_t off = 0;
                for (auto &c : ch) { CK(hipMemUnmap(base + off, c.second)); CK(hipMemMap(nb + off, c.second, 0, c.first, 0)); off += c.second; }
                if (mapped) CK(hipMemSetAccess(nb, mapped, &ad, 1));
                CK(hipMemAddressFree(base, reserved));
                base = nb; reserved = want; nreloc++;
            }
            hipMemGenericAllocationHandle_t h;
            CK(hipMemCreate(&h, grow, &prop, 0));
            CK(hipMemMap(base + mapped, grow, 0, h, 0));
            if (mode == 0) CK(hipMemSetAccess(base, mapped + grow, &ad, 1));
            else CK(hipMemSetAccess(base + mapped, grow, &ad, 1));
            ch.push_back({h, grow});
            // write the new chunk while a kernel that reads the old part may still be running
            if (mapped) hipLaunchKernelGGL(k_check, dim3(256), dim3(256), 0, nullptr, (const unsigned *)base, mapped / 4, 1000u * arena, bad);
            hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, nullptr, (unsigned *)(base + mapped), grow / 4, 1000u * arena + (unsigned)(mapped / 4));
            mapped += grow;
        }
        CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(k_check, dim3(1024), dim3(256), 0, nullptr, (const unsigned *)base, mapped / 4, 1000u * arena, bad);
        CK(hipDeviceSynchronize());
        int hb[4] = {0, 0, 0, 0}; CK(hipMemcpy(hb, bad, 16, hipMemcpyDeviceToHost)); CK(hipMemset(bad, 0, 16));
        printf("arena %d: %zu MB in %zu chunks, %d relocations, mismatches %d (first: got %u want %u at MB %d)\n", arena, mapped / MB, ch.size(), nreloc, hb[0], (unsigned)hb[1], (unsigned)hb[2], hb[3]);
        size_t off = 0;
        if (keep != 1) {
            for (auto &c : ch) { CK(hipMemUnmap(base + off, c.second)); CK(hipMemRelease(c.first)); off += c.second; }
            CK(hipMemAddressFree(base, reserved));
            if (keep == 2) { struct timespec ts = {3, 0}; nanosleep(&ts, nullptr); }
        }
    }
    printf("STRESS_DONE mode %d\n", mode);
    return 0;
}
