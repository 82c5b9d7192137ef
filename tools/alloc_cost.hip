// how expensive are the allocations of the placement walk?  (results: profiles/r06_placement_vmm.txt section 1; build: hipcc --offload-arch=gfx950 -O2 tools/alloc_cost.hip -o tools/bin/alloc_cost)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipFree(0);
    for (size_t gb : {1, 4, 16, 32}) {
        void *p = nullptr;
        double t0 = now();
        hipError_t e = hipMalloc(&p, gb << 30);
        double t1 = now();
        hipFree(p);
        double t2 = now();
        printf("hipMalloc %2zu GB: %8.1f ms (%s), hipFree %8.1f ms\n", gb, t1 - t0, hipGetErrorString(e), t2 - t1);
    }
    // second round: warm?
    for (size_t gb : {16, 16}) {
        void *p = nullptr;
        double t0 = now();
        hipMalloc(&p, gb << 30);
        double t1 = now();
        hipFree(p);
        printf("again hipMalloc %2zu GB: %8.1f ms, free %.1f\n", gb, t1 - t0, now() - t1);
    }
    // VMM: physical allocation without a mapping
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
    printf("granularity %zu\n", gran);
    for (size_t gb : {1, 16, 32}) {
        hipMemGenericAllocationHandle_t h;
        double t0 = now();
        hipError_t e = hipMemCreate(&h, gb << 30, &prop, 0);
        double t1 = now();
        printf("hipMemCreate %2zu GB: %8.1f ms (%s)", gb, t1 - t0, hipGetErrorString(e));
        if (e == hipSuccess) {
            void *va = nullptr;
            double t2 = now();
            e = hipMemAddressReserve(&va, gb << 30, 0, nullptr, 0);
            hipError_t e2 = hipMemMap(va, gb << 30, 0, h, 0);
            hipMemAccessDesc d = {};
            d.location = prop.location;
            d.flags = hipMemAccessFlagsProtReadWrite;
            hipError_t e3 = hipMemSetAccess(va, gb << 30, &d, 1);
            double t3 = now();
            printf("  reserve+map+access %8.1f ms (%s %s %s)", t3 - t2, hipGetErrorString(e), hipGetErrorString(e2), hipGetErrorString(e3));
            // partial map?
            hipMemUnmap(va, gb << 30);
            double t4 = now();
            hipError_t e4 = hipMemMap(va, 1ull << 30, 0, h, 0);
            printf("  partial map of 1 GB: %s", hipGetErrorString(e4));
            if (e4 == hipSuccess) { hipMemSetAccess(va, 1ull << 30, &d, 1); printf(" %.1f ms", now() - t4); hipMemUnmap(va, 1ull << 30); }
            hipMemAddressFree(va, gb << 30);
            double t5 = now();
            hipMemRelease(h);
            printf("  release %.1f ms", now() - t5);
        }
        printf("\n");
    }
    return 0;
}
