// sketch_body.h -- the per-thread body of sketch_tiles_kernel (sketch.hip): 32 consecutive positions of a 64-base window
// -> one seed bit and one marker bit per position.  In a header of its own so that profiles/calib/sketch_body_bench.hip
// times exactly the code the kernel runs.
#pragma once
// The canonical 21-mer is taken with ONE v_min_f64 on 42-bit integers read as (denormal) doubles: correct only while f64 denormals
// are preserved.  A fast-math build would flush every marker k-mer to zero without any error.
#ifdef __FAST_MATH__
#error "sketch_body.h orders 42-bit integers as f64 denormals (v_min_f64): do not build with -ffast-math / -Ofast"
#endif
#include "common.h"

// mm_hash64 (include/skder_amd_spec.h): key = ~(key + (key << 21)); key ^= key >> 24; key *= 265; key ^= key >> 14; key *= 21;
// key ^= key >> 28; key += key << 31.  The complement of the first step is moved behind the first xor-shift, where it is one
// xor of the high word: with p = key + (key << 21), ~p ^ (~p >> 24) == p ^ (p >> 24) ^ 0xFFFFFF0000000000.
#define SK_M21 0x200001u        /* 2^21 + 1 */
#define SK_M31 0x80000001u      /* 2^31 + 1 */

// What the body is built from, one bit per choice (profiles/calib/sketch_body_bench.hip times every combination; the kernel
// runs SK_BODY_DEFAULT; 0 = everything left to the compiler, the parity reference of the others):
#define SKB_DERIVE15   1      /* the 15-mers are read off the 21-mer registers instead of being rolled separately */
#define SKB_MUL_SPLIT  2      /* x * c = v_mad_u64_u32(lo, c, 0) + v_mul_lo_u32(hi, c) added into the high half */
#define SKB_MUL21_LSHL 4      /* x * 21 = two v_lshl_add_u64 */
#define SKB_PUSH_CARRY 8      /* threshold test: v_cmp_gt_u64 + v_addc_co_u32 mask, mask, mask */
#define SKB_FIRST_STEP 16     /* first step by key width: one multiply-add (15-mer), + v_mad_u32_u24 for the 10-bit high word (21-mer) */
#define SKB_MULLO_C    32     /* with SKB_MUL_SPLIT: the high word's product in plain C (the compiler may re-fuse it) */
#define SKB_M31_LSHL   64     /* x * (2^31 + 1): v_mad_u64_u32(lo) and hi + (hi << 31) through v_lshl_add_u32 + v_add_u32 */
#define SKB_M21_LSHL   128    /* 21-mer's first step: v_mad_u64_u32(lo) and hi + (hi << 21) through v_lshl_add_u32 + v_add_u32 */
#define SKB_ASM_HASH   8192   /* both hashes of a position as ONE hand-written instruction stream on fixed scratch registers (overrides the multiply / push choices) */
#define SKB_MUL_2MAD   16384  /* x * c = v_mad_u64_u32(lo, c, 0), v_mad_u64_u32(hi, c, 0) and ONE add into the high half (no register moves) */
#define SKB_MIN_F64    32768  /* canonical 21-mer through v_min_f64: 42-bit integers order like the denormal doubles they are */
#define SKB_NO_SEEDS   256    /* measurement only: skip the 15-mer hash */
#define SKB_NO_MARKS   512    /* measurement only: skip the 21-mer hash */
#define SKB_X_NOCM     1024   /* measurement only: the forward 21-mer stands in for the canonical one */
#define SKB_X_NOCS     2048   /* measurement only: the forward 15-mer stands in for the canonical one */
#define SKB_X_NORM     4096   /* measurement only: the reverse register is not rolled (implies wrong masks) */
#ifndef SK_BODY_DEFAULT
#define SK_BODY_DEFAULT (SKB_DERIVE15 | SKB_PUSH_CARRY | SKB_MUL21_LSHL | SKB_MIN_F64)
#endif

__device__ __forceinline__ uint64_t mk64(uint32_t lo, uint32_t hi) { return ((uint64_t)hi << 32) | lo; }

// x * (2^n + 1) with the high word's share hi + (hi << n) formed without a multiplier
template <int N>
__device__ __forceinline__ uint64_t mul_pow2p1(uint64_t x)
{
    const uint64_t a = (uint64_t)(uint32_t)x * ((1u << N) + 1u);        // v_mad_u64_u32 lo, c, 0
    const uint32_t hi = (uint32_t)(x >> 32);
    uint32_t t;
    asm("v_lshl_add_u32 %0, %1, %2, %1" : "=v"(t) : "v"(hi), "n"(N));
    return mk64((uint32_t)a, (uint32_t)(a >> 32) + t);
}
template <int V>
__device__ __forceinline__ uint64_t mul_c(uint64_t x, uint32_t c)
{
    if (V & SKB_MUL_2MAD) {
        // both products opaque: left to itself the compiler folds the add into a multiply-add whose addend {0, hi * c} it
        // assembles with two register moves (or forms hi * c with the slow v_mul_lo_u32)
        uint64_t a, b, carry;
        asm("v_mad_u64_u32 %0, %2, %3, %5, 0\n\tv_mad_u64_u32 %1, %2, %4, %5, 0"
            : "=&v"(a), "=v"(b), "=&s"(carry) : "v"((uint32_t)x), "v"((uint32_t)(x >> 32)), "s"(c));
        return mk64((uint32_t)a, (uint32_t)(a >> 32) + (uint32_t)b);
    }
    if ((V & SKB_M31_LSHL) && c == SK_M31) return mul_pow2p1<31>(x);
    if (!(V & SKB_MUL_SPLIT)) return x * c;
    const uint64_t a = (uint64_t)(uint32_t)x * c;                       // v_mad_u64_u32 lo, c, 0
    uint32_t t;
    if (V & SKB_MULLO_C) t = (uint32_t)(x >> 32) * c;
    else asm("v_mul_lo_u32 %0, %1, %2" : "=v"(t) : "v"((uint32_t)(x >> 32)), "s"(c));   // opaque: the compiler would fuse it back into a second 64-bit multiply-add
    return mk64((uint32_t)a, (uint32_t)(a >> 32) + t);
}
template <int V>
__device__ __forceinline__ uint64_t mul_21(uint64_t x)
{
    if (!(V & SKB_MUL21_LSHL)) return mul_c<V>(x, 21u);
    uint64_t t, u;
    asm("v_lshl_add_u64 %0, %1, 2, %1" : "=v"(t) : "v"(x));            // 5 x
    asm("v_lshl_add_u64 %0, %1, 4, %2" : "=v"(u) : "v"(x), "v"(t));    // 16 x + 5 x
    return u;
}
// the hash behind its first step p = key * (2^21 + 1)
template <int V>
__device__ __forceinline__ uint64_t mm_hash64_tail(uint64_t p)
{
    p = p ^ (p >> 24) ^ 0xFFFFFF0000000000ull;
    p = mul_c<V>(p, 265u);
    p = p ^ (p >> 14);
    p = mul_21<V>(p);
    p = p ^ (p >> 28);
    return mul_c<V>(p, SK_M31);
}
// hash of a canonical k-mer of `bits` bits
template <int V, int BITS>
__device__ __forceinline__ uint64_t mm_hash64_kmer(uint64_t key)
{
    if ((V & SKB_M21_LSHL) && BITS > 32) return mm_hash64_tail<V>(mul_pow2p1<21>(key));
    if (!(V & SKB_FIRST_STEP)) return mm_hash64_tail<V>(mul_c<V>(key, SK_M21));
    const uint64_t a = (uint64_t)(uint32_t)key * SK_M21;
    if (BITS <= 32) return mm_hash64_tail<V>(a);                         // no high word
    uint32_t h;                                                          // high word below 2^24, constant below 2^24
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(h) : "v"((uint32_t)(key >> 32)), "s"(SK_M21), "v"((uint32_t)(a >> 32)));
    return mm_hash64_tail<V>(mk64((uint32_t)a, h));
}
// bit j of the mask <- (h < thr).  Carry form: mask = 2 * mask + (h < thr), position j ends up at bit 31 - j
template <int V>
__device__ __forceinline__ void push_below(uint32_t &mask, uint64_t h, uint64_t thr, int j)
{
    if (V & SKB_PUSH_CARRY) asm("v_cmp_gt_u64 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(mask) : "v"(h), "s"(thr) : "vcc");
    else if (h < thr) mask |= 1u << j;
}

// Both hashes of one position and their threshold bits as ONE hand-written instruction stream (SKB_ASM_HASH; NOT the
// default: measured 100 ns per position against 93 for the compiler-scheduled default, profiles/round3_sketch_body.json --
// an assembly block is issued strictly in order, and the compiler's own interleaving of the two chains with the rolling
// code around them does better; kept as a measured alternative).  Multiplications as two v_mad_u64_u32 and one add, x * 21
// as two v_lshl_add_u64, xor-shifts as v_lshrrev_b64 + two xors, the threshold bit through the carry.  Scratch registers
// are fixed (v56-v63: the kernel stays at 64 VGPRs = 8 wavefronts per SIMD) because inline assembly cannot name the halves
// of a 64-bit operand.
__device__ __forceinline__ void hash2_push(uint32_t cs, uint32_t cm_lo, uint32_t cm_hi, uint32_t &smask, uint32_t &mmask)
{
#define MUL2(P0, P1, T0, T1, C)                                      \
        "v_mad_u64_u32 v[" #T0 ":" #T1 "], %[c0], v" #P1 ", " C ", 0\n\t"   \
        "v_mad_u64_u32 v[" #P0 ":" #P1 "], %[c1], v" #P0 ", " C ", 0\n\t"   \
        "v_add_u32 v" #P1 ", v" #P1 ", v" #T0 "\n\t"
#define XSH(P0, P1, T0, T1, N)                                       \
        "v_lshrrev_b64 v[" #T0 ":" #T1 "], " #N ", v[" #P0 ":" #P1 "]\n\t" \
        "v_xor_b32 v" #P0 ", v" #P0 ", v" #T0 "\n\t"                      \
        "v_xor_b32 v" #P1 ", v" #P1 ", v" #T1 "\n\t"
#define MUL21(P0, P1, T0, T1)                                        \
        "v_lshl_add_u64 v[" #T0 ":" #T1 "], v[" #P0 ":" #P1 "], 2, v[" #P0 ":" #P1 "]\n\t" \
        "v_lshl_add_u64 v[" #P0 ":" #P1 "], v[" #P0 ":" #P1 "], 4, v[" #T0 ":" #T1 "]\n\t"
    uint64_t c0, c1;      // carry-outs nobody reads (not vcc: back-to-back writers of one SGPR pair wait for each other)
    asm("v_mad_u64_u32 v[56:57], %[c0], %[cs], %[m21], 0\n\t"      // 15-mer: p = cs * (2^21 + 1)
        "v_mad_u64_u32 v[62:63], %[c1], %[cmh], %[m21], 0\n\t"     // 21-mer: p = cm * (2^21 + 1)
        "v_mad_u64_u32 v[60:61], %[c0], %[cml], %[m21], 0\n\t"
        "v_add_u32 v61, v61, v62\n\t"
        "v_lshrrev_b64 v[58:59], 24, v[56:57]\n\t"                // p ^= p >> 24, ^ 0xFFFFFF00 on the high word
        "v_lshrrev_b64 v[62:63], 24, v[60:61]\n\t"
        "v_xor_b32 v56, v56, v58\n\t"
        "v_xor_b32 v60, v60, v62\n\t"
        "v_bitop3_b32 v57, v57, v59, %[cx] bitop3:0x96\n\t"
        "v_bitop3_b32 v61, v61, v63, %[cx] bitop3:0x96\n\t"
        MUL2(56, 57, 58, 59, "%[m265]") MUL2(60, 61, 62, 63, "%[m265]")      // p *= 265
        XSH(56, 57, 58, 59, 14) XSH(60, 61, 62, 63, 14)            // p ^= p >> 14
        MUL21(56, 57, 58, 59) MUL21(60, 61, 62, 63)                // p *= 21
        XSH(56, 57, 58, 59, 28) XSH(60, 61, 62, 63, 28)            // p ^= p >> 28
        MUL2(56, 57, 58, 59, "%[m31]") MUL2(60, 61, 62, 63, "%[m31]")      // p *= 2^31 + 1
        "v_cmp_gt_u64 vcc, %[ts], v[56:57]\n\t"                   // mask = 2 * mask + (hash < threshold)
        "v_addc_co_u32 %[sm], vcc, %[sm], %[sm], vcc\n\t"
        "v_cmp_gt_u64 vcc, %[tm], v[60:61]\n\t"
        "v_addc_co_u32 %[mm], vcc, %[mm], %[mm], vcc"
        : [sm] "+v"(smask), [mm] "+v"(mmask), [c0] "=&s"(c0), [c1] "=&s"(c1)
        : [cs] "v"(cs), [cml] "v"(cm_lo), [cmh] "v"(cm_hi), [m21] "s"(SK_M21), [cx] "s"(0xFFFFFF00u), [m265] "s"(265u), [m31] "s"(SK_M31),
          [ts] "s"((uint64_t)SK_SEED_THR), [tm] "s"((uint64_t)SK_MARK_THR)
        : "vcc", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63");
#undef MUL2
#undef XSH
#undef MUL21
}

// w0..w3: the thread's 64-base window, 2 bits per base, oldest base in the lowest pair of w0; positions = the bases of w2, w3.
// Bit j of smask / mmask: the canonical 15-mer / 21-mer ENDING at base 32 + j of the window is sampled.
template <int V>
__device__ __forceinline__ void sketch_body(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t &smask, uint32_t &mmask)
{
    smask = 0; mmask = 0;
    uint32_t fs = 0, rs = 0;        // separate 15-mer registers (unless derived)
    uint64_t fm = 0, rm = 0;        // 21-mer registers: fm newest base lowest; rm its reverse complement, newest base's complement on top
#pragma unroll
    for (int n = 12; n < 64; n++) {
        const uint32_t w = n < 16 ? w0 : (n < 32 ? w1 : (n < 48 ? w2 : w3));
        const uint32_t b = (w >> (2 * (n & 15))) & 3u;
        fm = ((fm << 2) | b) & SK_MARK_MASK;
        if (!(V & SKB_X_NORM)) rm = (rm >> 2) | ((uint64_t)(3u - b) << 40);
        if (!(V & SKB_DERIVE15)) {
            fs = ((fs << 2) | b) & SK_SEED_MASK;
            rs = (rs >> 2) | ((3u - b) << 28);
        }
        if (n < 32) continue;       // warm-up: the 20 bases in front of the thread's first position
        const int j = n - 32;
        if (V & SKB_DERIVE15) { fs = (uint32_t)fm & SK_SEED_MASK; rs = (uint32_t)(rm >> 12); }
        const uint32_t cs = (V & SKB_X_NOCS) ? fs : (fs < rs ? fs : rs);
        uint64_t cm;
        if (V & SKB_X_NOCM) cm = fm;
        else if (V & SKB_MIN_F64) {
            double m;
            asm("v_min_f64 %0, %1, %2" : "=v"(m) : "v"(__longlong_as_double((long long)fm)), "v"(__longlong_as_double((long long)rm)));
            cm = (uint64_t)__double_as_longlong(m);
        } else cm = fm < rm ? fm : rm;
        if (V & SKB_ASM_HASH) { hash2_push(cs, (uint32_t)cm, (uint32_t)(cm >> 32), smask, mmask); continue; }
        if (!(V & SKB_NO_SEEDS)) push_below<V>(smask, mm_hash64_kmer<V, 30>((uint64_t)cs), SK_SEED_THR, j);
        else smask ^= cs;                   // (measurement variants: keep the canonical forms alive)
        if (!(V & SKB_NO_MARKS)) push_below<V>(mmask, mm_hash64_kmer<V, 42>(cm), SK_MARK_THR, j);
        else mmask ^= (uint32_t)cm ^ (uint32_t)(cm >> 32);
    }
    if ((V & (SKB_PUSH_CARRY | SKB_ASM_HASH)) && !(V & SKB_NO_SEEDS)) smask = __brev(smask);
    if ((V & (SKB_PUSH_CARRY | SKB_ASM_HASH)) && !(V & SKB_NO_MARKS)) mmask = __brev(mmask);
}
