// scan.hip -- device-wide exclusive scan used by the compaction steps (tile counts, bucket
// counts, pair capacities).  2048 elements per workgroup; block sums scanned recursively.
#include "device_utils.h"

#define SCAN_ELEMS 8
#define SCAN_BLOCK (256 * SCAN_ELEMS)

__global__ __launch_bounds__(256) void scan_reduce_kernel(const uint32_t *__restrict__ in, uint32_t *__restrict__ bsum, size_t n)
{
    __shared__ uint32_t wsum[4];
    size_t base = (size_t)blockIdx.x * SCAN_BLOCK + (size_t)threadIdx.x * SCAN_ELEMS;
    uint32_t s = 0;
#pragma unroll
    for (int e = 0; e < SCAN_ELEMS; e++)
        if (base + e < n) s += in[base + e];
    uint32_t total;
    (void)block_excl_scan_256(s, wsum, total);
    if (threadIdx.x == 0) bsum[blockIdx.x] = total;
}

__global__ __launch_bounds__(256) void scan_apply_kernel(const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
                                                         const uint32_t *__restrict__ boff, size_t n)
{
    __shared__ uint32_t wsum[4];
    size_t base = (size_t)blockIdx.x * SCAN_BLOCK + (size_t)threadIdx.x * SCAN_ELEMS;
    uint32_t v[SCAN_ELEMS];
    uint32_t s = 0;
#pragma unroll
    for (int e = 0; e < SCAN_ELEMS; e++) {
        v[e] = (base + e < n) ? in[base + e] : 0u;
        s += v[e];
    }
    uint32_t total;
    uint32_t ex = block_excl_scan_256(s, wsum, total) + (boff ? boff[blockIdx.x] : 0u);
#pragma unroll
    for (int e = 0; e < SCAN_ELEMS; e++) {
        if (base + e < n) out[base + e] = ex;
        ex += v[e];
    }
}

void exclusive_scan_u32(const uint32_t *d_in, uint32_t *d_out, size_t n, ScanWorkspace &ws, hipStream_t st, int level)
{
    if (n == 0) return;
    size_t nb = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    if (nb == 1) {
        hipLaunchKernelGGL(scan_apply_kernel, dim3(1), dim3(256), 0, st, d_in, d_out, (const uint32_t *)nullptr, n);
        return;
    }
    while ((int)ws.levels.size() <= level) ws.levels.push_back(new DevBuf<uint32_t>());
    DevBuf<uint32_t> &b = *ws.levels[level];
    b.resize(2 * nb, st);
    uint32_t *bsum = b.p, *boff = b.p + nb;
    hipLaunchKernelGGL(scan_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, st, d_in, bsum, n);
    exclusive_scan_u32(bsum, boff, nb, ws, st, level + 1);
    hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)nb), dim3(256), 0, st, d_in, d_out, (const uint32_t *)boff, n);
}
