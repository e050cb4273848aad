// repro: which VMM sequences does this HIP runtime accept?  (hipcc --offload-arch=gfx950 scratch/vmm_repro.hip -o scratch/vmm_repro)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
static bool ok_all = true;
#define T(x) do { hipError_t e_ = (x); printf("  %-70s %s\n", #x, hipGetErrorString(e_)); if (e_ != hipSuccess) { (void)hipGetLastError(); ok_all = false; } } while (0)
#define TOUCH(x) do { if (ok_all) { T(x); } else printf("  (skipped: %s)\n", #x); } while (0)
int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    (void)hipSetDevice(0); (void)hipFree(nullptr);
    hipMemAllocationProp prop{}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc ad{}; ad.location = prop.location; ad.flags = hipMemAccessFlagsProtReadWrite;
    const size_t MB = 1 << 20;
    size_t g1 = 0, g2 = 0;
    (void)hipMemGetAllocationGranularity(&g1, &prop, hipMemAllocationGranularityMinimum);
    (void)hipMemGetAllocationGranularity(&g2, &prop, hipMemAllocationGranularityRecommended);
    printf("granularity min %zu recommended %zu\n", g1, g2);
    {
        printf("A: one range of 64 MB, chunks of 2 MB then 24 MB, access per chunk\n");
        void *b = nullptr; hipMemGenericAllocationHandle_t h1, h2;
        T(hipMemAddressReserve(&b, 64 * MB, 2 * MB, nullptr, 0));
        T(hipMemCreate(&h1, 2 * MB, &prop, 0)); T(hipMemMap(b, 2 * MB, 0, h1, 0)); T(hipMemSetAccess(b, 2 * MB, &ad, 1));
        T(hipMemCreate(&h2, 24 * MB, &prop, 0)); T(hipMemMap((char *)b + 2 * MB, 24 * MB, 0, h2, 0)); T(hipMemSetAccess((char *)b + 2 * MB, 24 * MB, &ad, 1));
        printf("   ... the same access call over the whole mapped range instead:\n");
        T(hipMemSetAccess(b, 26 * MB, &ad, 1));
        TOUCH(hipMemset(b, 0, 26 * MB)); TOUCH(hipDeviceSynchronize()); ok_all = true;
    }
    {
        printf("B: one range of 64 MB, chunks of 24 MB then 2 MB\n");
        void *b = nullptr; hipMemGenericAllocationHandle_t h1, h2;
        T(hipMemAddressReserve(&b, 64 * MB, 2 * MB, nullptr, 0));
        T(hipMemCreate(&h1, 24 * MB, &prop, 0)); T(hipMemMap(b, 24 * MB, 0, h1, 0)); T(hipMemSetAccess(b, 24 * MB, &ad, 1));
        T(hipMemCreate(&h2, 2 * MB, &prop, 0)); T(hipMemMap((char *)b + 24 * MB, 2 * MB, 0, h2, 0)); T(hipMemSetAccess((char *)b + 24 * MB, 2 * MB, &ad, 1));
        TOUCH(hipMemset(b, 0, 26 * MB)); TOUCH(hipDeviceSynchronize()); ok_all = true;
    }
    {
        printf("C: map everything first, then ONE access call for the whole range\n");
        void *b = nullptr; hipMemGenericAllocationHandle_t h1, h2;
        T(hipMemAddressReserve(&b, 64 * MB, 2 * MB, nullptr, 0));
        T(hipMemCreate(&h1, 2 * MB, &prop, 0)); T(hipMemMap(b, 2 * MB, 0, h1, 0));
        T(hipMemCreate(&h2, 24 * MB, &prop, 0)); T(hipMemMap((char *)b + 2 * MB, 24 * MB, 0, h2, 0));
        T(hipMemSetAccess(b, 26 * MB, &ad, 1));
        TOUCH(hipMemset(b, 0, 26 * MB)); TOUCH(hipDeviceSynchronize()); ok_all = true;
    }
    {
        printf("D: alignment 0 in the reservation (as scratch/alloc_cost.hip), 2 MB then 24 MB\n");
        void *b = nullptr; hipMemGenericAllocationHandle_t h1, h2;
        T(hipMemAddressReserve(&b, 64 * MB, 0, nullptr, 0));
        T(hipMemCreate(&h1, 2 * MB, &prop, 0)); T(hipMemMap(b, 2 * MB, 0, h1, 0)); T(hipMemSetAccess(b, 2 * MB, &ad, 1));
        T(hipMemCreate(&h2, 24 * MB, &prop, 0)); T(hipMemMap((char *)b + 2 * MB, 24 * MB, 0, h2, 0)); T(hipMemSetAccess((char *)b + 2 * MB, 24 * MB, &ad, 1));
        TOUCH(hipMemset(b, 0, 26 * MB)); TOUCH(hipDeviceSynchronize()); ok_all = true;
    }
    {
        ok_all = true;
        printf("E: equal chunks of 8 MB x 3 (as the micro-benchmark did)\n");
        void *b = nullptr; hipMemGenericAllocationHandle_t h[3];
        T(hipMemAddressReserve(&b, 64 * MB, 2 * MB, nullptr, 0));
        for (int i = 0; i < 3; i++) { T(hipMemCreate(&h[i], 8 * MB, &prop, 0)); T(hipMemMap((char *)b + i * 8 * MB, 8 * MB, 0, h[i], 0)); T(hipMemSetAccess((char *)b + i * 8 * MB, 8 * MB, &ad, 1)); }
    }
    {
        ok_all = true;
        printf("F: relocation: handle mapped at one range, unmapped, mapped at another; then a second chunk behind it\n");
        void *b = nullptr, *nb = nullptr; hipMemGenericAllocationHandle_t h1, h2;
        T(hipMemAddressReserve(&b, 2 * MB, 2 * MB, nullptr, 0));
        T(hipMemCreate(&h1, 2 * MB, &prop, 0)); T(hipMemMap(b, 2 * MB, 0, h1, 0)); T(hipMemSetAccess(b, 2 * MB, &ad, 1));
        T(hipMemset(b, 7, 2 * MB)); T(hipDeviceSynchronize());
        T(hipMemAddressReserve(&nb, 26 * MB, 2 * MB, nullptr, 0));
        T(hipMemUnmap(b, 2 * MB)); T(hipMemMap(nb, 2 * MB, 0, h1, 0)); T(hipMemSetAccess(nb, 2 * MB, &ad, 1));
        T(hipMemAddressFree(b, 2 * MB));
        T(hipMemCreate(&h2, 24 * MB, &prop, 0)); T(hipMemMap((char *)nb + 2 * MB, 24 * MB, 0, h2, 0)); T(hipMemSetAccess((char *)nb + 2 * MB, 24 * MB, &ad, 1));
        unsigned char v = 0; TOUCH(hipMemcpy(&v, nb, 1, hipMemcpyDeviceToHost)); printf("   first byte after the move: %d (7 = contents travelled)\n", v);
    }
    return 0;
}
