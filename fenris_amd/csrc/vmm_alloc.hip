// Device allocations through the virtual-memory API (hipMemAddressReserve / hipMemCreate / hipMemMap) with an explicit physical chunk
// size: the one lever a caller has over how a large `values` array is backed (result: profiles/r05_vmm_experiment.txt).
// No reference counterpart.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "../../include/fenris_hip.h"

namespace {
struct VmmAlloc {
    void* base = nullptr;
    size_t reserved = 0;
    std::vector<hipMemGenericAllocationHandle_t> handles;
    std::vector<size_t> sizes;
};
std::mutex g_mu;
std::unordered_map<void*, VmmAlloc> g_allocs;

void release(VmmAlloc& a) {
    size_t off = 0;
    for (size_t k = 0; k < a.handles.size(); ++k) {
        (void)hipMemUnmap(static_cast<char*>(a.base) + off, a.sizes[k]);
        (void)hipMemRelease(a.handles[k]);
        off += a.sizes[k];
    }
    if (a.base) (void)hipMemAddressFree(a.base, a.reserved);
}
}  // namespace

extern "C" {

// `bytes` of device memory on `device`, mapped from physical chunks of `chunk_bytes` each (rounded up to the allocation granularity;
// 0: one chunk for the whole array; the last chunk may be shorter).  *granularity_out (may be NULL) receives the granularity used.
int fh_vmm_alloc(int device, uint64_t bytes, uint64_t chunk_bytes, void** out, uint64_t* granularity_out) {
    if (!out || bytes == 0) return FH_BAD_ARGUMENT;
    *out = nullptr;
    int ndev = 0, prev = -1;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return FH_BAD_ARGUMENT;
    // the reservation and the mappings are made with `device` current, and the caller's device is restored on every way out
    (void)hipGetDevice(&prev);
    struct Restore { int d; ~Restore() { if (d >= 0) (void)hipSetDevice(d); } } restore{prev};
    if (hipSetDevice(device) != hipSuccess) return FH_HIP_ERROR;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || gran == 0) return FH_HIP_ERROR;
    if (granularity_out) *granularity_out = gran;
    auto round_up = [&](size_t x) { return (x + gran - 1) / gran * gran; };
    const size_t total = round_up((size_t)bytes);
    const size_t chunk = chunk_bytes ? round_up((size_t)chunk_bytes) : total;
    VmmAlloc a;
    a.reserved = total;
    if (hipMemAddressReserve(&a.base, total, 0, nullptr, 0) != hipSuccess) return FH_HIP_ERROR;
    size_t off = 0;
    while (off < total) {
        const size_t sz = std::min(chunk, total - off);
        hipMemGenericAllocationHandle_t h;
        if (hipMemCreate(&h, sz, &prop, 0) != hipSuccess) { release(a); return FH_HIP_ERROR; }
        if (hipMemMap(static_cast<char*>(a.base) + off, sz, 0, h, 0) != hipSuccess) { (void)hipMemRelease(h); release(a); return FH_HIP_ERROR; }
        a.handles.push_back(h);
        a.sizes.push_back(sz);
        off += sz;
    }
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = device;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if (hipMemSetAccess(a.base, total, &acc, 1) != hipSuccess) { release(a); return FH_HIP_ERROR; }
    *out = a.base;
    std::lock_guard<std::mutex> lk(g_mu);
    g_allocs.emplace(a.base, std::move(a));
    return FH_OK;
}

int fh_vmm_free(void* ptr) {
    VmmAlloc a;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_allocs.find(ptr);
        if (it == g_allocs.end()) return FH_BAD_ARGUMENT;
        a = std::move(it->second);
        g_allocs.erase(it);
    }
    release(a);
    return FH_OK;
}

}  // extern "C"
