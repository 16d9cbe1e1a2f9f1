// profiles/calib/sketch_body_bench.hip -- issue cost of the body of sketch_tiles_kernel (skder_amd/csrc/sketch_body.h) for
// every combination of its instruction-selection choices.  Each thread runs the 32-position body on ROUNDS pseudo-random
// 64-base windows held in registers (no memory traffic beyond one store at the end), 8 wavefronts per SIMD on every CU --
// the kernel's own occupancy --, so the time is what the body costs to ISSUE.  Every variant must produce the same masks
// (checksum compared with variant 0, the compiler's selection).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../skder_amd/csrc -o sketch_body_bench sketch_body_bench.hip && ./sketch_body_bench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "sketch_body.h"

#define ROUNDS 64

template <int V>
__global__ __launch_bounds__(256) void body_kernel(uint64_t *out, uint32_t seed)
{
    uint32_t x = seed ^ (blockIdx.x * 256u + threadIdx.x) * 0x9E3779B9u;
    uint64_t acc = 0;
    for (int r = 0; r < ROUNDS; r++) {
        uint32_t w[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; w[k] = x; }
        uint32_t sm, mm;
        sketch_body<V>(w[0], w[1], w[2], w[3], sm, mm);
        acc = acc * 0x100000001B3ull + (((uint64_t)mm << 32) | sm);
    }
    out[blockIdx.x * 256u + threadIdx.x] = acc;
}

typedef void (*kern_t)(uint64_t *, uint32_t);
struct Var { int v; const char *what; kern_t k; };
#define VAR(v, what) {v, what, body_kernel<v>}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount * 8 * 4;      // 8 workgroups of 4 wavefronts per CU resident, 4 such sets
    const size_t n = (size_t)blocks * 256;
    uint64_t *d;
    hipMalloc(&d, n * 8);
    std::vector<uint64_t> h(n);
#define RO (SKB_DERIVE15 | SKB_NO_SEEDS | SKB_NO_MARKS)
#define B13 (SKB_DERIVE15 | SKB_PUSH_CARRY | SKB_MUL21_LSHL)
    const Var vars[] = {
        VAR(0, "compiler's selection"),
        VAR(SKB_DERIVE15, "15-mers derived from the 21-mer registers"),
        VAR(SKB_MUL_SPLIT, "x*c = mad(lo) + mul_lo(hi) + add"),
        VAR(SKB_MUL21_LSHL, "x*21 = two v_lshl_add_u64"),
        VAR(SKB_PUSH_CARRY, "threshold bit through the carry"),
        VAR(SKB_FIRST_STEP, "first step by key width (v_mad_u32_u24)"),
        VAR(SKB_M31_LSHL, "x*(2^31+1): high share by v_lshl_add_u32"),
        VAR(SKB_M21_LSHL, "21-mer first step: high share by v_lshl_add_u32"),
        VAR(SKB_MIN_F64, "canonical 21-mer by v_min_f64"),
        VAR(SKB_DERIVE15 | SKB_PUSH_CARRY, "derive + carry"),
        VAR(B13, "derive + carry + lshl21"),
        VAR(B13 | SKB_MIN_F64, "derive + carry + lshl21 + min_f64 (SK_BODY_DEFAULT)"),
        VAR(B13 | SKB_MUL_2MAD | SKB_MIN_F64, "derive + carry + lshl21 + min_f64 + two opaque multiply-adds and an add"),
        VAR(SKB_DERIVE15 | SKB_MUL_SPLIT | SKB_MUL21_LSHL | SKB_PUSH_CARRY | SKB_FIRST_STEP, "round-3 first attempt (mul_lo, mad_u32_u24)"),
        VAR(SKB_ASM_HASH | SKB_DERIVE15 | SKB_MIN_F64, "both hashes as one hand-written stream + derived 15-mers + min_f64"),
        VAR(SKB_NO_SEEDS, "compiler's, 21-mer hash only"),
        VAR(SKB_NO_MARKS, "compiler's, 15-mer hash only"),
        VAR(SKB_NO_SEEDS | SKB_NO_MARKS, "compiler's, rolling + canonical forms only"),
        VAR(RO, "derived, rolling + canonical forms only"),
        VAR(RO | SKB_MIN_F64, "derived, rolling + canonical forms only, min_f64"),
        VAR(RO | SKB_X_NOCM, "  ... without the 64-bit min"),
        VAR(RO | SKB_X_NOCS, "  ... without the 32-bit min"),
        VAR(RO | SKB_X_NOCM | SKB_X_NOCS, "  ... without both"),
    };
    uint64_t want = 0;
    printf("{\"device\": \"%s\", \"cus\": %d, \"positions_per_thread\": %d, \"variants\": [", p.gcnArchName, p.multiProcessorCount, 32 * ROUNDS);
    for (size_t i = 0; i < sizeof(vars) / sizeof(vars[0]); i++) {
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(vars[i].k, dim3(blocks), dim3(256), 0, 0, d, 12345u);
        hipDeviceSynchronize();
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(a, 0);
            hipLaunchKernelGGL(vars[i].k, dim3(blocks), dim3(256), 0, 0, d, 12345u);
            hipEventRecord(b, 0);
            hipEventSynchronize(b);
            float ms = 0;
            hipEventElapsedTime(&ms, a, b);
            best = ms < best ? ms : best;
        }
        hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
        uint64_t sum = 0;
        for (size_t k = 0; k < n; k++) sum = sum * 31 + h[k];
        const bool measure_only = vars[i].v & (SKB_NO_SEEDS | SKB_NO_MARKS | SKB_X_NOCM | SKB_X_NOCS | SKB_X_NORM);
        if (i == 0) want = sum;
        const double positions = (double)n * 32 * ROUNDS;
        // ns per position-wavefront per SIMD: time * SIMDs / (positions / 64)
        const double ns = best * 1e6 * p.multiProcessorCount * 4 / (positions / 64.0);
        printf("%s{\"variant\": %d, \"what\": \"%s\", \"ms\": %.3f, \"ns_per_position_wave_per_simd\": %.2f, \"Gpos_per_s\": %.1f, \"masks\": \"%s\"}", i ? ", " : "",
               vars[i].v, vars[i].what, best, ns, positions / best / 1e6, measure_only ? "n/a" : (sum == want ? "equal" : "DIFFERENT"));
    }
    printf("]}\n");
    return 0;
}
