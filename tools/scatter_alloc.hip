// tools/scatter_alloc.hip - round 4 probe: a virtually contiguous device buffer whose PHYSICAL chunks (hipMemCreate, `chunk`
// bytes each) are mapped in a shuffled order - does a page-scattered layout change how the single-launch kernels run?
// (development aid; nothing of the product links against it)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>

extern "C" size_t scat_granularity(int recommended) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t g = 0;
    if (hipMemGetAllocationGranularity(&g, &prop, recommended ? hipMemAllocationGranularityRecommended : hipMemAllocationGranularityMinimum) != hipSuccess) return 0;
    return g;
}

// order: 0 = chunks mapped in creation order, 1 = shuffled (seed), 2 = reversed
extern "C" int scat_alloc(size_t bytes, size_t chunk, int order, unsigned seed, void** out) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    const size_t nch = (bytes + chunk - 1) / chunk;
    void* va = nullptr;
    hipError_t e = hipMemAddressReserve(&va, nch * chunk, chunk, nullptr, 0);
    if (e != hipSuccess) return (int)e;
    std::vector<hipMemGenericAllocationHandle_t> h(nch);
    for (size_t i = 0; i < nch; ++i) {
        e = hipMemCreate(&h[i], chunk, &prop, 0);
        if (e != hipSuccess) return 1000 + (int)e;
    }
    std::vector<size_t> perm(nch);
    for (size_t i = 0; i < nch; ++i) perm[i] = order == 2 ? nch - 1 - i : i;
    if (order == 1) {
        srand(seed);
        for (size_t i = nch - 1; i > 0; --i) { const size_t j = (size_t)rand() % (i + 1); const size_t t = perm[i]; perm[i] = perm[j]; perm[j] = t; }
    }
    for (size_t i = 0; i < nch; ++i) {
        e = hipMemMap(reinterpret_cast<char*>(va) + i * chunk, chunk, 0, h[perm[i]], 0);
        if (e != hipSuccess) return 2000 + (int)e;
    }
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = dev;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    e = hipMemSetAccess(va, nch * chunk, &acc, 1);
    if (e != hipSuccess) return 3000 + (int)e;
    for (size_t i = 0; i < nch; ++i) (void)hipMemRelease(h[i]);      // the mappings keep the memory alive
    *out = va;
    return 0;
}

// The SAME physical chunks mapped several times: `nviews` virtual ranges over one set of `chunk`-byte allocations, view v in
// the order given by perm[v * nch + i] (physical chunk of virtual chunk i).  out[v] receives the views' addresses.
extern "C" int scat_views(size_t bytes, size_t chunk, int nviews, const uint32_t* perm, void** out) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    const size_t nch = (bytes + chunk - 1) / chunk;
    std::vector<hipMemGenericAllocationHandle_t> h(nch);
    for (size_t i = 0; i < nch; ++i) {
        const hipError_t e = hipMemCreate(&h[i], chunk, &prop, 0);
        if (e != hipSuccess) return 1000 + (int)e;
    }
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = dev;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    for (int v = 0; v < nviews; ++v) {
        void* va = nullptr;
        hipError_t e = hipMemAddressReserve(&va, nch * chunk, chunk, nullptr, 0);
        if (e != hipSuccess) return 4000 + (int)e;
        for (size_t i = 0; i < nch; ++i) {
            e = hipMemMap(reinterpret_cast<char*>(va) + i * chunk, chunk, 0, h[perm[(size_t)v * nch + i]], 0);
            if (e != hipSuccess) return 2000 + (int)e;
        }
        e = hipMemSetAccess(va, nch * chunk, &acc, 1);
        if (e != hipSuccess) return 3000 + (int)e;
        out[v] = va;
    }
    for (size_t i = 0; i < nch; ++i) (void)hipMemRelease(h[i]);
    return 0;
}
