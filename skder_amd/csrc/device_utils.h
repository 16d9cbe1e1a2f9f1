// device_utils.h -- wave64 / workgroup primitives and a device-wide exclusive scan.
#pragma once
#include "common.h"

// exclusive scan of one value per lane across a 64-wide wavefront; total = sum over the wave
__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, uint32_t &total)
{
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t y = __shfl_up(x, o, 64);
        if (lane >= (uint32_t)o) x += y;
    }
    total = __shfl(x, 63, 64);
    return x - v;
}

// exclusive scan across a 256-thread workgroup (4 waves). wsum: __shared__ uint32_t[4].
// Safe to call repeatedly: ends with a barrier-protected read of wsum.
__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t v, uint32_t *wsum, uint32_t &total)
{
    uint32_t wt;
    uint32_t ex = wave_excl_scan(v, wt);
    const uint32_t wave = threadIdx.x >> 6;
    __syncthreads();                 // previous users of wsum are done
    if ((threadIdx.x & 63u) == 63u) wsum[wave] = wt;
    __syncthreads();
    uint32_t s0 = wsum[0], s1 = wsum[1], s2 = wsum[2], s3 = wsum[3];
    total = s0 + s1 + s2 + s3;
    uint32_t off = (wave > 0 ? s0 : 0) + (wave > 1 ? s1 : 0) + (wave > 2 ? s2 : 0);
    return ex + off;
}

// exclusive scan across a workgroup of NW waves (NW <= 16). wsum: __shared__ uint32_t[NW].
template <int NW>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *wsum, uint32_t &total)
{
    uint32_t wt;
    uint32_t ex = wave_excl_scan(v, wt);
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    __syncthreads();                 // previous users of wsum are done
    if (lane == 63u) wsum[wave] = wt;
    __syncthreads();
    // every wave scans the NW wave totals itself (NW <= 64 lanes)
    uint32_t t = lane < (uint32_t)NW ? wsum[lane] : 0u, tt;
    const uint32_t tex = wave_excl_scan(t, tt);
    total = tt;
    return ex + __shfl(tex, (int)wave, 64);
}

// ---------------------------------------------------------------------------------------------
// device-wide exclusive scan (u32), three-pass, recursive on the block sums

struct ScanWorkspace {
    std::vector<DevBuf<uint32_t> *> levels;
    ~ScanWorkspace() { for (auto *b : levels) delete b; }
};

void exclusive_scan_u32(const uint32_t *d_in, uint32_t *d_out, size_t n, ScanWorkspace &ws, hipStream_t st, int level = 0);
