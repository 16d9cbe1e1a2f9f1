// profiles/calib/valu_rates.hip -- issue cost of the integer VALU instructions the sketch kernel is made of,
// relative to v_xor_b32, on gfx950.  Every kernel runs the same loop: 8 independent register chains, 32 rounds
// unrolled, so nothing waits on a result; 8 waves per SIMD on every CU.  Output: ns per wave-instruction per
// SIMD and the ratio to v_xor_b32.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define ROUNDS 32
#define ITERS 2000

#define KERNEL32(name, text)                                                                      \
    __global__ void name(uint32_t *out, uint32_t seed)                                              \
    {                                                                                               \
        uint32_t r[8];                                                                              \
        for (int i = 0; i < 8; i++) r[i] = seed * (threadIdx.x + i + 1);                            \
        uint32_t c = seed | 3u;                                                                     \
        for (int it = 0; it < ITERS; it++) {                                                        \
            _Pragma("unroll") for (int k = 0; k < ROUNDS; k++) {                                    \
                _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile(text : "+v"(r[i]) : "v"(c) : "vcc"); \
            }                                                                                       \
        }                                                                                           \
        uint32_t s = 0;                                                                             \
        for (int i = 0; i < 8; i++) s ^= r[i];                                                      \
        if (s == 0x12345678u) out[threadIdx.x] = s;                                                 \
    }

#define KERNEL64(name, text)                                                                      \
    __global__ void name(uint32_t *out, uint32_t seed)                                              \
    {                                                                                               \
        uint64_t r[8];                                                                              \
        for (int i = 0; i < 8; i++) r[i] = (uint64_t)seed * (threadIdx.x + i + 1) * 0x9E3779B97F4A7C15ull; \
        uint32_t c = seed | 3u;                                                                     \
        uint64_t c64 = ((uint64_t)seed << 32) | 5u;                                                 \
        for (int it = 0; it < ITERS; it++) {                                                        \
            _Pragma("unroll") for (int k = 0; k < ROUNDS; k++) {                                    \
                _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile(text : "+v"(r[i]) : "v"(c), "v"(c64) : "vcc"); \
            }                                                                                       \
        }                                                                                           \
        uint64_t s = 0;                                                                             \
        for (int i = 0; i < 8; i++) s ^= r[i];                                                      \
        if (s == 0x12345678u) out[threadIdx.x] = (uint32_t)s;                                       \
    }

KERNEL32(k_xor, "v_xor_b32 %0, %0, %1")
KERNEL32(k_add, "v_add_u32 %0, %0, %1")
KERNEL32(k_lshl, "v_lshlrev_b32 %0, 3, %0")
KERNEL32(k_alignbit, "v_alignbit_b32 %0, %0, %1, 7")
KERNEL32(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
KERNEL32(k_mul_hi, "v_mul_hi_u32 %0, %0, %1")
KERNEL32(k_mad24, "v_mad_u32_u24 %0, %0, %1, %0")
KERNEL32(k_bitop3, "v_bitop3_b32 %0, %0, %1, %0 bitop3:0x96")
KERNEL32(k_lshl_add32, "v_lshl_add_u32 %0, %0, 3, %1")
KERNEL32(k_xad, "v_xad_u32 %0, %0, %1, %0")
KERNEL64(k_lshl64, "v_lshlrev_b64 %0, 3, %0")
KERNEL64(k_lshr64, "v_lshrrev_b64 %0, 3, %0")
KERNEL64(k_lshl_add64, "v_lshl_add_u64 %0, %0, 3, %2")
KERNEL64(k_mad64, "v_mad_u64_u32 %0, vcc, %1, %1, %0")
KERNEL64(k_cmp64, "v_cmp_lt_u64 vcc, %0, %2")
KERNEL32(k_add_co, "v_add_co_u32 %0, vcc, %0, %1")
KERNEL32(k_and, "v_and_b32 %0, %0, %1")
KERNEL32(k_or, "v_or_b32 %0, %0, %1")
KERNEL32(k_not, "v_not_b32 %0, %0")
KERNEL32(k_mov, "v_mov_b32 %0, %1")
KERNEL32(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL32(k_min, "v_min_u32 %0, %0, %1")
KERNEL32(k_sub, "v_sub_u32 %0, %0, %1")
KERNEL32(k_lshr, "v_lshrrev_b32 %0, 3, %0")
KERNEL32(k_or3, "v_or3_b32 %0, %0, %1, %0")
KERNEL32(k_and_or, "v_and_or_b32 %0, %0, %1, %0")
KERNEL32(k_lshl_or, "v_lshl_or_b32 %0, %0, 3, %1")
KERNEL32(k_bfe, "v_bfe_u32 %0, %0, 3, 20")
KERNEL32(k_add3, "v_add3_u32 %0, %0, %1, %0")
KERNEL32(k_perm, "v_perm_b32 %0, %0, %1, %1")
KERNEL32(k_cmp32, "v_cmp_lt_u32 vcc, %0, %1")
KERNEL32(k_addc, "v_addc_co_u32 %0, vcc, %0, %1, vcc")
KERNEL32(k_mul24, "v_mul_u32_u24 %0, %0, %1")
KERNEL32(k_xor_e64, "v_xor_b32_e64 %0, %0, %1")
KERNEL32(k_bfrev, "v_bfrev_b32 %0, %0")
KERNEL32(k_pk_add, "v_pk_add_u16 %0, %0, %1")

// scalar ALU, and a 1:1 mix of scalar and vector instructions of one wave
__global__ void k_salu(uint32_t *out, uint32_t seed)
{
    uint32_t r[8];
    for (int i = 0; i < 8; i++) r[i] = __builtin_amdgcn_readfirstlane(seed * (i + 1));
    uint32_t c = __builtin_amdgcn_readfirstlane(seed | 3u);
    for (int it = 0; it < ITERS; it++) {
        _Pragma("unroll") for (int k = 0; k < ROUNDS; k++) {
            _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile("s_add_u32 %0, %0, %1" : "+s"(r[i]) : "s"(c) : "scc");
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s ^= r[i];
    if (s == 0x12345678u) out[threadIdx.x] = s;
}
__global__ void k_mix(uint32_t *out, uint32_t seed)     // ROUNDS x 8 pairs (one scalar + one vector instruction)
{
    uint32_t r[8], v[8];
    for (int i = 0; i < 8; i++) { r[i] = __builtin_amdgcn_readfirstlane(seed * (i + 1)); v[i] = seed * (threadIdx.x + i + 1); }
    uint32_t c = __builtin_amdgcn_readfirstlane(seed | 3u), cv = seed | 5u;
    for (int it = 0; it < ITERS; it++) {
        _Pragma("unroll") for (int k = 0; k < ROUNDS; k++) {
            _Pragma("unroll") for (int i = 0; i < 8; i++) {
                asm volatile("s_add_u32 %0, %0, %1" : "+s"(r[i]) : "s"(c) : "scc");
                asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[i]) : "v"(cv));
            }
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s ^= r[i] ^ v[i];
    if (s == 0x12345678u) out[threadIdx.x] = s;
}

typedef void (*kern_t)(uint32_t *, uint32_t);

static double run(kern_t k, uint32_t *d, int blocks, int threads = 512)
{
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, 7u);
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, 7u);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    int cus = p.multiProcessorCount;
    int blocks = cus * 4;   // 4 x 512 threads = 32 waves per CU = 8 per SIMD
    uint32_t *d;
    hipMalloc(&d, 4096);
    struct { const char *name; kern_t k; } ks[] = {
        {"v_xor_b32", k_xor}, {"v_add_u32", k_add}, {"v_lshlrev_b32", k_lshl}, {"v_alignbit_b32", k_alignbit},
        {"v_mul_lo_u32", k_mul_lo}, {"v_mul_hi_u32", k_mul_hi}, {"v_mad_u32_u24", k_mad24}, {"v_bitop3_b32", k_bitop3},
        {"v_lshl_add_u32", k_lshl_add32}, {"v_xad_u32", k_xad}, {"v_lshlrev_b64", k_lshl64}, {"v_lshrrev_b64", k_lshr64},
        {"v_lshl_add_u64", k_lshl_add64}, {"v_mad_u64_u32", k_mad64}, {"v_cmp_lt_u64", k_cmp64}, {"v_add_co_u32", k_add_co},
        {"v_and_b32", k_and}, {"v_or_b32", k_or}, {"v_not_b32", k_not}, {"v_mov_b32", k_mov}, {"v_cndmask_b32", k_cndmask},
        {"v_min_u32", k_min}, {"v_sub_u32", k_sub}, {"v_lshrrev_b32", k_lshr}, {"v_or3_b32", k_or3}, {"v_and_or_b32", k_and_or},
        {"v_lshl_or_b32", k_lshl_or}, {"v_bfe_u32", k_bfe}, {"v_add3_u32", k_add3}, {"v_perm_b32", k_perm}, {"v_cmp_lt_u32", k_cmp32},
        {"v_addc_co_u32", k_addc}, {"v_mul_u32_u24", k_mul24}, {"v_xor_b32_e64", k_xor_e64}, {"v_bfrev_b32", k_bfrev}, {"v_pk_add_u16", k_pk_add},
    };
    double base = 0;
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"rates\": {", p.gcnArchName, cus, p.clockRate / 1000);
    for (size_t i = 0; i < sizeof(ks) / sizeof(ks[0]); i++) {
        double ms = run(ks[i].k, d, blocks);
        // wave-instructions per SIMD: 8 waves x ITERS x ROUNDS x 8
        double per = ms * 1e6 / (8.0 * ITERS * ROUNDS * 8);
        if (i == 0) base = per;
        // one wave per SIMD: the 8 chains of that single wave are all the independence there is
        double lat = run(ks[i].k, d, cus, 256) * 1e6 / ((double)ITERS * ROUNDS * 8);
        printf("%s\"%s\": {\"ns_per_wave_instr\": %.3f, \"vs_xor\": %.2f, \"one_wave_ns\": %.3f}", i ? ", " : "", ks[i].name, per, per / base, lat);
    }
    // how the issue rate grows with the number of resident waves per SIMD (256-thread workgroups: one wave per SIMD each)
    struct { const char *name; kern_t k; double per_round; } ws[] = {
        {"v_xor_b32", k_xor, 1}, {"v_lshlrev_b32", k_lshl, 1}, {"v_mad_u64_u32", k_mad64, 1}, {"s_add_u32", k_salu, 1}, {"s_add_u32+v_xor_b32", k_mix, 2},
    };
    printf("}, \"instr_per_cycle_per_simd_by_waves\": {");
    const int nw[] = {1, 2, 3, 4, 6, 8};
    for (size_t i = 0; i < sizeof(ws) / sizeof(ws[0]); i++) {
        printf("%s\"%s\": {", i ? ", " : "", ws[i].name);
        for (size_t j = 0; j < sizeof(nw) / sizeof(nw[0]); j++) {
            double ms = run(ws[i].k, d, cus * nw[j], 256);
            double instr = (double)nw[j] * ITERS * ROUNDS * 8 * ws[i].per_round;      // wave-instructions per SIMD
            printf("%s\"%d\": %.3f", j ? ", " : "", nw[j], instr / (ms * 1e-3 * 2.4e9));
        }
        printf("}");
    }
    printf("}}\n");
    return 0;
}
