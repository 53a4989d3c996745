// Host staging memory of the set-up stages (round 5).
//
// The stages that run on the host -- chains of owner blocks, the merge of the lane-table hashes, the lane tuner -- move arrays of 1 - 25 MB
// between device and host.  As std::vector they were mapped anew for every context (first-touch page faults: ~1 us per 4 KB page) and, worse,
// RETURNED to the operating system when the stage ended: glibc unmaps a block of that size on free, and an unmap of memory that a device copy
// has touched makes the kernel driver suspend the process's GPU queues until it has revalidated its mappings.  Measured on the 216^3 mesh
// (general Hex8 path): the first k_hex8_rows launch started 27 ms after it was enqueued; with MALLOC_MMAP_MAX_=0 / MALLOC_TRIM_THRESHOLD_
// set (nothing ever unmapped) it started at once (profiles/r05_setup.txt).  So these arrays come from a process-wide pool of power-of-two
// blocks that is never given back while the process lives (bounded: beyond POOL_LIMIT bytes retained a block is freed after all).
#pragma once
#include <cstddef>
#include <new>
#include <vector>

namespace fenris_hip {

struct HostPool {
    static constexpr size_t SMALL = 64 * 1024;                    // below: plain malloc / free (the heap, nothing is unmapped)
    static constexpr size_t POOL_LIMIT = (size_t)1 << 30;         // bytes retained at most (the set-up of the 216^3 mesh stages < 200 MB at a time)
    static void* take(size_t bytes);
    static void give(void* p, size_t bytes) noexcept;
    static size_t retained_bytes();                               // (tests)
    static size_t trim();                                         // frees every retained block (fh_host_pool_trim); returns the bytes freed
};

template <typename T>
struct PoolAllocator {
    typedef T value_type;
    PoolAllocator() noexcept {}
    template <typename U>
    PoolAllocator(const PoolAllocator<U>&) noexcept {}
    T* allocate(size_t n) {
        void* p = HostPool::take(n * sizeof(T));
        if (!p) throw std::bad_alloc();
        return static_cast<T*>(p);
    }
    void deallocate(T* p, size_t n) noexcept { HostPool::give(p, n * sizeof(T)); }
    template <typename U>
    bool operator==(const PoolAllocator<U>&) const noexcept { return true; }
    template <typename U>
    bool operator!=(const PoolAllocator<U>&) const noexcept { return false; }
};

// vector of the set-up stages: storage from the pool
template <typename T>
using hvec = std::vector<T, PoolAllocator<T>>;

// array without initialisation (a device copy fills it)
template <typename T>
struct HostBuf {
    T* p = nullptr;
    size_t n = 0;
    HostBuf() = default;
    explicit HostBuf(size_t count) { alloc(count); }
    HostBuf(const HostBuf&) = delete;
    HostBuf& operator=(const HostBuf&) = delete;
    ~HostBuf() { release(); }
    void release() noexcept {
        if (p) HostPool::give(p, n * sizeof(T));
        p = nullptr;
        n = 0;
    }
    void alloc(size_t count) {
        release();
        if (count == 0) count = 1;
        p = static_cast<T*>(HostPool::take(count * sizeof(T)));
        if (!p) throw std::bad_alloc();
        n = count;
    }
    T* data() { return p; }
    const T* data() const { return p; }
    size_t size() const { return n; }
    T& operator[](size_t i) { return p[i]; }
    const T& operator[](size_t i) const { return p[i]; }
};

}  // namespace fenris_hip
