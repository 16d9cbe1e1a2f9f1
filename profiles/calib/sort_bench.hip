// sort_bench.hip -- how long does a stable device-wide sort of the step's 120 M seeds by their 30-bit k-mer take?
// (the price of a multi-genome seed index: DESIGN "what would come next" (1)).  rocPRIM radix sort, key u32, value u64.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <cstring>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void fill(uint32_t *k, uint64_t *v, size_t n)
{
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t x = i * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    k[i] = (uint32_t)x & 0x3FFFFFFFu; v[i] = i;
}
int main(int argc, char **argv)
{
    const size_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 120000000ull;
    uint32_t *k0, *k1; uint64_t *v0, *v1;
    CK(hipMalloc(&k0, n * 4)); CK(hipMalloc(&k1, n * 4)); CK(hipMalloc(&v0, n * 8)); CK(hipMalloc(&v1, n * 8));
    fill<<<(unsigned)((n + 255) / 256), 256>>>(k0, v0, n);
    size_t tmp = 0;
    CK(rocprim::radix_sort_pairs(nullptr, tmp, k0, k1, v0, v1, n, 0, 30));
    void *d_tmp; CK(hipMalloc(&d_tmp, tmp));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int bits : {30, 24, 16}) for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(a));
        CK(rocprim::radix_sort_pairs(d_tmp, tmp, k0, k1, v0, v1, n, 0, bits));
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("n %zu, %d key bits, u32 key + u64 value: %.2f ms (%.1f G pairs/s), temporary %.1f MB\n", n, bits, ms, n / ms / 1e6, tmp / 1e6);
    }
    // keys only (u64: k-mer << 32 | genome-local payload would not fit; for the rate)
    CK(rocprim::radix_sort_keys(nullptr, tmp, k0, k1, n, 0, 30));
    void *d_tmp2; CK(hipMalloc(&d_tmp2, tmp));
    for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(a)); CK(rocprim::radix_sort_keys(d_tmp2, tmp, k0, k1, n, 0, 30)); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("n %zu, 30 key bits, u32 keys only: %.2f ms\n", n, ms);
    }
    return 0;
}
