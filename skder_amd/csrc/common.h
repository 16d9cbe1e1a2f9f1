// common.h -- shared declarations of the MI355X ANI engine (libskder_amd.so).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/skder_amd.h"
#include "../../include/skder_amd_spec.h"

#define SK_SEED_MASK 0x3FFFFFFFu            /* 2*ANI_K bits */
#define SK_MARK_MASK ((1ULL << 42) - 1)      /* 2*ANI_MARKER_K bits */
#define SK_FWD_BIT 0x80000000u
#define SK_SEED_THR (0xFFFFFFFFFFFFFFFFULL / ANI_C)
#define SK_MARK_THR (0xFFFFFFFFFFFFFFFFULL / ANI_MARKER_C)

#define SK_THREADS 256
#define SK_POS_PER_THREAD 32
static_assert(SK_THREADS * SK_POS_PER_THREAD == SKDER_TILE, "tile geometry");
#define SK_SLOT_SEEDS 512     /* per-tile seed slot capacity (mean 65.5) */
#define SK_SLOT_MARKS 128     /* per-tile marker slot capacity (mean 8.2) */

struct SkError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

#define HIPCHECK(expr)                                                                           \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            throw SkError(std::string(#expr) + ": " + hipGetErrorString(e_) + " (" + __FILE__ + ":" + \
                          std::to_string(__LINE__) + ")");                                       \
    } while (0)

// one 8192-position slice of one kept FASTA record
struct TileDesc {
    uint64_t base_off;   // offset in d_bases of the tile's first position (multiple of 32)
    uint32_t genome;     // index inside the batch
    uint32_t ctg;        // kept-record index inside the genome
    uint32_t pos0;       // record-relative position of the tile start
    uint32_t npos;       // positions in the tile (<= SKDER_TILE)
    uint32_t gpos0;      // genome-relative gpos of the tile start
    uint32_t pad;
};

// caching device allocator: blocks freed by DevBuf are kept (per device) and handed out again, so a
// steady-state call sequence performs no hipMalloc/hipFree (both synchronise the device).
void *pool_alloc(size_t bytes);
void pool_free(void *p);
void pool_trim();   // return every cached block to the driver

// device buffer with geometric growth
template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0, cap = 0;
    ~DevBuf() { release(); }
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    void release() {
        if (p) pool_free(p);
        p = nullptr; n = cap = 0;
    }
    // make room for `want` elements; keeps the first `keep` elements
    void reserve(size_t want, size_t keep, hipStream_t st) {
        if (want <= cap) return;
        size_t nc = cap ? cap + cap / 2 : want;
        if (nc < want) nc = want;
        nc = (nc + 1023) & ~(size_t)1023;
        T *q = static_cast<T *>(pool_alloc(nc * sizeof(T)));
        if (keep) HIPCHECK(hipMemcpyAsync(q, p, keep * sizeof(T), hipMemcpyDeviceToDevice, st));
        if (p) { if (keep) HIPCHECK(hipStreamSynchronize(st)); pool_free(p); }
        p = q; cap = nc;
    }
    void resize(size_t want, hipStream_t st) { reserve(want, n, st); n = want; }
};

struct skder_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;   // second queue: the seed index is built here while the marker screen runs on `stream`
    std::string last_error;
    hipEvent_t ev[16];
    double timing[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    double timing_join = 0;
    double timing_runs = 0;          // ms of the run-extraction kernel in the last call
    double timing_index = 0;         // ms of the last index build (device)
    std::vector<skder_edge_t> edges;
    std::vector<uint32_t> pairs_ref, pairs_query;   // candidate pairs of the last screen_rows call
    uint32_t *d_flags = nullptr;   // [0] overflow / error flags from kernels
    uint64_t counters[4] = {0, 0, 0, 0};   // [0] chunks processed, [1] chunks sent to the slow path
    bool chain_attr_set = false;           // large-LDS opt-in of the join / finalize kernels done on this context's device
    void *chain_work = nullptr;            // grow-only work buffers of chain_pairs (chain.hip)
    void (*chain_work_free)(void *) = nullptr;
};
