// see host_pool.hpp
#include "host_pool.hpp"

#include <cstdlib>
#include <mutex>

namespace fenris_hip {

namespace {
struct Pool {
    std::mutex mu;
    std::vector<void*> free_blocks[48];   // by log2 of the block size
    size_t retained = 0;
};
Pool& pool() {
    static Pool* p = new Pool;   // never destroyed: blocks may be given back by static destructors of other units
    return *p;
}
int class_of(size_t bytes) {
    int k = 16;
    while (((size_t)1 << k) < bytes) ++k;
    return k;
}
}  // namespace

void* HostPool::take(size_t bytes) {
    if (bytes == 0) bytes = 1;
    if (bytes < SMALL) return std::malloc(bytes);
    const int k = class_of(bytes);
    if (k >= 48) return nullptr;
    {
        Pool& P = pool();
        std::lock_guard<std::mutex> g(P.mu);
        if (!P.free_blocks[k].empty()) {
            void* p = P.free_blocks[k].back();
            P.free_blocks[k].pop_back();
            P.retained -= (size_t)1 << k;
            return p;
        }
    }
    return std::malloc((size_t)1 << k);
}

void HostPool::give(void* p, size_t bytes) noexcept {
    if (!p) return;
    if (bytes == 0) bytes = 1;
    if (bytes < SMALL) { std::free(p); return; }
    const int k = class_of(bytes);
    Pool& P = pool();
    {
        std::lock_guard<std::mutex> g(P.mu);
        if (P.retained + ((size_t)1 << k) <= POOL_LIMIT) {
            try {
                P.free_blocks[k].push_back(p);
                P.retained += (size_t)1 << k;
                return;
            } catch (...) {
            }
        }
    }
    std::free(p);
}

size_t HostPool::trim() {
    Pool& P = pool();
    std::vector<void*> blocks;
    size_t freed = 0;
    {
        std::lock_guard<std::mutex> g(P.mu);
        for (auto& v : P.free_blocks) {
            blocks.insert(blocks.end(), v.begin(), v.end());
            v.clear();
        }
        freed = P.retained;
        P.retained = 0;
    }
    for (void* p : blocks) std::free(p);   // (outside the lock: unmapping touched memory can take tens of milliseconds, host_pool.hpp)
    return freed;
}

size_t HostPool::retained_bytes() {
    Pool& P = pool();
    std::lock_guard<std::mutex> g(P.mu);
    return P.retained;
}

}  // namespace fenris_hip
