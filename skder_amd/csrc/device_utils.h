// device_utils.h -- wave64 / workgroup primitives and a device-wide exclusive scan.
#pragma once
#include "common.h"

// exclusive scan of one value per lane across a 64-wide wavefront; total = sum over the wave
// Exclusive prefix sum over the 64 lanes of a wavefront.  Six DPP adds (row_shr 1 / 2 / 4 / 8 inside each row of 16 lanes, then
// row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3): no trip through the LDS crossbar, where six __shfl_up steps
// (ds_bpermute_b32 each, with a compare and a select) cost the run extraction 1.3 ms per step of the benchmark (round 6).
__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, uint32_t &total)
{
    uint32_t x = v;
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true);    // row_shr:1 (lanes without a source add 0)
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, true);    // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, true);    // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, true);    // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1, 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2, 3
    total = (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
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
