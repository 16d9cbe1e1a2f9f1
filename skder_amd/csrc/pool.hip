// pool.hip -- caching device allocator behind DevBuf (see common.h)
#include <map>
#include <mutex>
#include <thread>
#include <unordered_map>

#include "common.h"

namespace {
// A cached block may still have work pending on the stream of the context that freed it.  One host thread drives one
// context at a time (streams of one context are ordered by its own events), so a block handed to the thread that freed
// it needs nothing; a block that changes threads -- two contexts on one device used from different host threads, e.g. the
// per-device workers of the multi-device entry points -- is handed over only after the device has drained.
struct Cached { void *p; std::thread::id by; };
struct Pool {
    std::mutex mu;
    std::multimap<size_t, Cached> free_blocks;          // size -> block
    std::unordered_map<void *, size_t> size_of;          // every live or cached block
};
Pool &pool_for_current_device()
{
    constexpr int MAX_DEVICES = 64;       // a node holds 8 MI355X; partitioned (CPX) modes show up to 64 logical devices
    static Pool pools[MAX_DEVICES];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= MAX_DEVICES) throw SkError("device index " + std::to_string(dev) + " is beyond the allocator's " + std::to_string(MAX_DEVICES) + " per-device pools");
    return pools[dev];
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
            const Cached c = it->second;
            P.free_blocks.erase(it);
            if (c.by != std::this_thread::get_id()) (void)hipDeviceSynchronize();
            return c.p;
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
    Pool *pp = nullptr;
    try { pp = &pool_for_current_device(); } catch (...) {}      // called from destructors: never throws
    if (!pp) { (void)hipFree(p); return; }
    Pool &P = *pp;
    std::lock_guard<std::mutex> lk(P.mu);
    auto it = P.size_of.find(p);
    if (it == P.size_of.end()) { (void)hipFree(p); return; }
    P.free_blocks.emplace(it->second, Cached{p, std::this_thread::get_id()});
}

void pool_trim()
{
    Pool &P = pool_for_current_device();
    std::lock_guard<std::mutex> lk(P.mu);
    for (auto &kv : P.free_blocks) { (void)hipFree(kv.second.p); P.size_of.erase(kv.second.p); }
    P.free_blocks.clear();
}
