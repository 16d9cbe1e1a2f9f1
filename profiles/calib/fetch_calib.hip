// fetch_calib.hip -- calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of
// this library's kernels (MI355X_MICROARCH.md, HBM section: "calibrate on a known byte count in your own
// access pattern").  Every kernel moves a known number of bytes through a buffer far larger than the
// 256 MiB Infinity Cache; profiles/collect.sh runs it under `rocprofv3 --pmc FETCH_SIZE` and
// `--pmc WRITE_SIZE` and profiles/summarise.py turns counter / known bytes into correction factors.
//   A read16_coalesced   16 B per lane, consecutive lanes consecutive (sketch_tiles_kernel's base stream)
//   B read4_coalesced     4 B per lane, consecutive (join_probe_kernel's k-mer stream)
//   C read_line_per_lane  every lane reads its own 64-byte line with 4 x 16 B, lines 640 B apart
//                         (chain_fast_kernel's two streams); all bytes of the buffer are read once
//   D write4_coalesced    4 B per lane, consecutive (join_probe_kernel's hit words)
//   E gather4_random      4 B per lane at pseudo-random addresses in a 96 KB window that moves along the
//                         buffer (join_probe_kernel's position gather: window = one genome's sgpos array)
//   F gather16_random     16 B per lane at pseudo-random 16-byte slots of a 384 KB window that moves along the buffer
//                         (index_genome_lds_kernel's dominant read stream: one genome's (k-mer, position, record) records
//                         gathered through the bucket permutation; round 6)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void read16_coalesced(const uint4 *__restrict__ p, size_t n16, uint32_t *sink)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void read4_coalesced(const uint32_t *__restrict__ p, size_t n4, uint32_t *sink)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) acc ^= p[i];
    if (acc == 0x12345678u) *sink = acc;
}
// lane l of the grid owns a 640-byte strip (10 lines of 64 B) and walks it line by line, 4 x dwordx4 per line
__global__ void read_line_per_lane(const uint4 *__restrict__ p, size_t nstrips, uint32_t *sink)
{
    uint32_t acc = 0;
    for (size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x; s < nstrips; s += (size_t)gridDim.x * blockDim.x) {
        const uint4 *q = p + s * 40;   // 640 B = 40 x 16 B
        for (int line = 0; line < 10; line++) {
            const uint4 a = q[4 * line], b = q[4 * line + 1], c = q[4 * line + 2], d = q[4 * line + 3];
            acc ^= a.x ^ b.y ^ c.z ^ d.w;
        }
    }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void write4_coalesced(uint32_t *__restrict__ p, size_t n4)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i;
}
__global__ void gather4_random(const uint32_t *__restrict__ p, size_t n4, uint32_t *sink)
{
    uint32_t acc = 0;
    const size_t win = 24576;   // 96 KB of u32
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const size_t base = (i / win) * win;                      // every window is gathered from `win` times in total
        uint32_t h = (uint32_t)i * 2654435761u;
        h ^= h >> 15;
        const size_t j = base + h % win;
        acc ^= p[j < n4 ? j : 0];
    }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void gather16_random(const uint4 *__restrict__ p, size_t n16, uint32_t *sink)
{
    uint32_t acc = 0;
    const size_t win = 24576;   // 384 KB of 16-byte records: one 3 Mb genome's seeds
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const size_t base = (i / win) * win;
        uint32_t h = (uint32_t)i * 2654435761u;
        h ^= h >> 15;
        const size_t j = base + h % win;
        const uint4 v = p[j < n16 ? j : 0];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}

int main()
{
    const size_t bytes = 6ull << 30;   // 6 GiB >> 256 MiB Infinity Cache
    void *buf;
    uint32_t *sink;
    CK(hipMalloc(&buf, bytes));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(buf, 1, bytes));
    CK(hipDeviceSynchronize());
    const dim3 grid(256 * 16), block(256);
    hipLaunchKernelGGL(read16_coalesced, grid, block, 0, 0, (const uint4 *)buf, bytes / 16, sink);
    hipLaunchKernelGGL(read4_coalesced, grid, block, 0, 0, (const uint32_t *)buf, bytes / 4, sink);
    hipLaunchKernelGGL(read_line_per_lane, grid, block, 0, 0, (const uint4 *)buf, bytes / 640, sink);
    hipLaunchKernelGGL(write4_coalesced, grid, block, 0, 0, (uint32_t *)buf, bytes / 4);
    hipLaunchKernelGGL(gather4_random, grid, block, 0, 0, (const uint32_t *)buf, bytes / 4, sink);
    hipLaunchKernelGGL(gather16_random, grid, block, 0, 0, (const uint4 *)buf, bytes / 16, sink);
    CK(hipDeviceSynchronize());
    printf("known_bytes read16_coalesced %zu\nknown_bytes read4_coalesced %zu\nknown_bytes read_line_per_lane %zu\n"
           "known_bytes write4_coalesced %zu\nknown_bytes gather4_random %zu\nknown_bytes gather16_random %zu\n",
           bytes, bytes, (bytes / 640) * 640, bytes, bytes, bytes);
    return 0;
}
