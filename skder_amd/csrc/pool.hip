// pool.hip -- caching device allocator behind DevBuf (see common.h)
#include <map>
#include <mutex>
#include <unordered_map>

#include "common.h"

namespace {
struct Pool {
    std::mutex mu;
    std::multimap<size_t, void *> free_blocks;          // size -> block
    std::unordered_map<void *, size_t> size_of;          // every live or cached block
};
Pool &pool_for_current_device()
{
    static Pool pools[16];
    int dev = 0;
    (void)hipGetDevice(&dev);
    return pools[dev & 15];
}
}   // namespace

void *pool_alloc(size_t bytes)
{
    if (bytes == 0) bytes = 256;
    bytes = (bytes + 255) & ~(size_t)255;
    Pool &P = pool_for_current_device();
    {
        std::lock_guard<std::mutex> lk(P.mu);
        auto it = P.free_blocks.lower_bound(bytes);
        if (it != P.free_blocks.end() && it->first <= bytes + bytes / 2 + (1u << 20)) {
            void *p = it->second;
            P.free_blocks.erase(it);
            return p;
        }
    }
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
        pool_trim();
        e = hipMalloc(&p, bytes);
        if (e != hipSuccess) throw SkError(std::string("hipMalloc of ") + std::to_string(bytes) + " bytes failed: " + hipGetErrorString(e));
    }
    std::lock_guard<std::mutex> lk(P.mu);
    P.size_of[p] = bytes;
    return p;
}

void pool_free(void *p)
{
    if (!p) return;
    Pool &P = pool_for_current_device();
    std::lock_guard<std::mutex> lk(P.mu);
    auto it = P.size_of.find(p);
    if (it == P.size_of.end()) { (void)hipFree(p); return; }
    P.free_blocks.emplace(it->second, p);
}

void pool_trim()
{
    Pool &P = pool_for_current_device();
    std::lock_guard<std::mutex> lk(P.mu);
    for (auto &kv : P.free_blocks) { (void)hipFree(kv.second); P.size_of.erase(kv.second); }
    P.free_blocks.clear();
}
