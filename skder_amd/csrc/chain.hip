// chain.hip -- anchors, chunked chaining, ANI / aligned fraction for a list of genome pairs.
//
// Device restatement of oracle/ani_oracle.c oracle_pair() (steps 1-6); integer results are
// bit-identical by construction, the few double operations are + - * / in the oracle's order
// (this file is compiled with -ffp-contract=off).
//
//   join_probe_kernel  hit word of every (pair, seed of the chunked genome), the probed genome's index in LDS
//   run_extract_kernel seed-parallel: runs of seeds that continue the previous hit -> run records per 256-seed segment
//   chain_runs_kernel  one LANE per (pair, 20 kb chunk): banded DP over the chunk's run records against a
//                      4-run register ring; exact for "simple" chunks, which it proves as it goes;
//                      everything else is handed to the slow path.
//   slow_*_kernel      unabridged algorithm for the declined chunks: ordered anchors via the bucket
//                      index (one wave per chunk), full band-50 DP and best-first chain extraction
//                      with back-tracking (one lane per chunk).
//   finalize_kernel  one workgroup per pair: chains into LDS, better-chain overlap filter as a
//                    parallel fix-point, fixed-point containment ANI, aligned fraction.
#include <algorithm>
#include <chrono>
#include <thread>
#include <type_traits>

#include "device_utils.h"
#include "engine.h"
#include "screen.h"

struct SetView {
    const GenomeMeta *meta;
    const uint32_t *pkmer, *pgpos, *pchunk;   // position order
    const uint8_t *pcs;                       // position order: 1 = first seed of its chunk
    const uint32_t *skmer, *sgpos, *sctg;     // bucket order
    const uint32_t *stag;                     // bucket order: sgpos | (sctg & 63) << 24 | strand of the k-mer << 31
    const uint32_t *boff;
    const uint32_t *chunk_start;
    const uint32_t *rec_goff;
};

struct PairDesc {
    uint32_t q, r;          // chunked genome, other genome (indices inside their sets)
    uint32_t chunk_base;    // first work item (chunk) of this pair in the batch
    uint32_t n_chunks;
    uint32_t c_base, c_cap; // chain-record region of the slow path
    uint32_t flags;         // bit0: chunked genome is the pair's Query; bit1: q in set B; bit2: r in set B; bit3: every chunk takes the slow path
    uint32_t hit_base;      // first entry of this pair in the hit array (one u32 per seed of the chunked genome)
    uint32_t multi_base, multi_cap;   // region of 4-hit records for seeds with several hits
    uint32_t rec_base, rec_cap;   // region of the pair's run records
    uint32_t q_chunk_off;         // offset of the chunked genome's chunk table (GenomeMeta::chunk_off)
    uint32_t seg_per, seg_a;      // run extraction: 256-seed segments per quarter of the pair; seed_off & 3 of the chunked genome
    uint32_t pad;                 // 64 bytes: one cache line per descriptor
};

// hit[s] for seed s of the chunked genome: gpos on the other genome | rev<<31, or one of
#define HIT_NONE 0xFFFFFFFFu      // no occurrence on the other genome
#define HIT_MULTI 0x7F000000u     // | slot: 2..4 occurrences, listed (ascending gpos) in multi[slot]
#define HIT_MANY 0x7FFFFFFFu      // more than 4 occurrences (or no room): the chunk takes the slow path

struct ChainRec {
    int32_t score;
    uint32_t n, n_seeds, q0, q1, r0, r1, chunk;   // chunk: index of the chain's 20 kb cell inside its pair
};

struct PairOut {
    uint64_t cell_seeds;               // all seeds of the chunked genome in the cells that hold a kept chain
    uint64_t sum_seeds, sum_anchors, sum_span;
    uint32_t n_chains, n_chains_all, n_anchors, pad;
    double ani_raw, ani_span, ani, af_q, af_r;   // q = chunked genome
};

#define USED_BIT 0x80000000u
#define FIN_LDS_CHAINS 2048
#define FIN_BINS 1024
#define FAST_SLOTS 3
#ifndef CF_OCC
#define CF_OCC 3           // wavefronts per SIMD chain_fast_kernel is compiled for
#endif
#ifndef PASS_THRESH
#define PASS_THRESH 8u       // parked lanes of a wavefront that start a general pass of chain_fast_kernel
#endif
#define RING 4
#define CHUNK_SLOW 0xFFFFFFFFu
#define SUCC_BIT 0x80000000u

__device__ __forceinline__ uint32_t find_pair(const PairDesc *__restrict__ pairs, uint32_t npairs, uint32_t t)
{
    uint32_t lo = 0, hi = npairs;
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (pairs[mid].chunk_base <= t) lo = mid; else hi = mid;
    }
    return lo;
}

// ---------------------------------------------------------------------------------------------
// JOIN: hit words for every (pair, seed of the chunked genome), R-stationary.
// Pairs are sorted by the probed genome R.  One 1024-thread workgroup takes a group of (at most 8) pairs that
// share R, loads R's bucket offsets and, per bucket-ordered seed, the REMAINDER of its k-mer (the
// 30 - bits bits of the mixed k-mer that the bucket number does not fix: 16 bits for genomes of
// 16 k seeds and more) into LDS once -- 80 KB for a 3 Mb genome, so two workgroups share a CU --
// and then streams the position-ordered k-mers of every chunked genome of the group past it: one
// coalesced 4-byte read per seed, a probe of the LDS-resident bucket (about 4 LDS reads), a gather of
// the matched position (with its record tag and strand) from R's stag array (L2-resident), and one
// coalesced 4-byte hit word written per seed -- in position order, so nothing is scattered into HBM
// and no memset is needed.  Genomes whose index does not fit in LDS are probed in several passes over
// bucket ranges.
// Bucket offsets take ONE BYTE per bucket in LDS: four buckets share a 32-bit group word -- the offset of the group's first
// seed (16 bits, relative to the pass) and the four bucket sizes (4 bits each; 15 = "15 or more": that bucket's bounds are read
// from the global table) -- so a probe reads one word where it used to read two 16-bit offsets, and the table of a 3 Mb genome
// (16 K buckets, 25 K remainders) is 66 KB instead of 83: TWO workgroups per CU up to 4 Mb.  (Round 2's layout fitted two only
// up to 24.5 K seeds -- the benchmark's genomes have 23-25 K, and the kernel's 101 scalar registers admitted one workgroup per
// CU whatever the LDS said: it ran at half the wavefronts it was designed for.)
struct JoinGroup { uint32_t pair_begin, pair_end; };
#define JOIN_THREADS 1024
#define JOIN_U 4             // seeds per thread and trip
#ifndef JOIN_PROBE_N
#define JOIN_PROBE_N 4       // bucket entries compared without a loop
#endif
#define JOIN_SLACK 64u       // readable entries behind the last remainder (the unconditional 8-entry compare; a group with an oversize bucket)
#define JOIN_SMEM_MAX (155u * 1024u)   // dynamic LDS of a workgroup at most
#define JOIN_SMEM_TWO (80u * 1024u)    // up to here two workgroups fit a CU

// LDS bytes wanted for a whole-table pass over a genome with 2^bits buckets and n seeds
static inline size_t join_need(uint32_t bits, uint32_t n)
{
    return (size_t)(1u << bits) + 64u + ((size_t)n + JOIN_SLACK) * (bits >= 14u ? 2u : 4u);      // one byte per bucket (group words), remainders
}

// the probe loop of one staged bucket range for all pairs of a group.  FP: remainder type (16 bits once
// the genome has 2^14 buckets, else 32); FITS: the range's remainders are in LDS (false only for a
// single bucket with more seeds than LDS holds); WHOLE: the range is the whole table, so every seed
// belongs to this pass (the normal case: both true, no per-seed tests for either)
template <typename FP, bool FITS, bool WHOLE>
__device__ __forceinline__ void join_pass(const SetView &A, const SetView &B, const PairDesc *__restrict__ pairs, const JoinGroup g,
                                          uint32_t *__restrict__ hits, uint4 *__restrict__ multi, uint32_t *__restrict__ pair_nmulti,
                                          const FP *s_fp, const uint32_t *s_grp, const uint32_t *__restrict__ rb, const uint32_t *__restrict__ rk,
                                          const uint32_t *__restrict__ rg, uint32_t base, uint32_t bits, uint32_t bb0, uint32_t bb1,
                                          uint32_t rrep, uint32_t tid)
{
    const uint32_t bsh = 30u - bits, rmask = (1u << bsh) - 1u;
    for (uint32_t p = g.pair_begin; p < g.pair_end; p++) {
        const PairDesc pd = pairs[p];
        const SetView &QS = (pd.flags & 2u) ? B : A;
        const GenomeMeta *Qm = QS.meta + pd.q;
        const uint32_t *pk = QS.pkmer + Qm->seed_off;
        const uint32_t nq = Qm->n_seeds;
        uint32_t *hit = hits + pd.hit_base;
        // JOIN_U independent seeds per thread and trip, handled in phases so that the memory operations
        // of all of them are in flight together: k-mer loads, LDS probes, then ALL position gathers, then
        // the coalesced hit-word stores; the rare multi-occurrence seeds come last
        // FULL trips (every seed of every thread exists) carry no bounds tests; one guarded trip finishes the pair
        auto trip = [&](auto full_tag, const uint32_t s0) {
            constexpr bool FULL = decltype(full_tag)::value;
            uint32_t kqv[JOIN_U], remv[JOIN_U], lov[JOIN_U], hiv[JOIN_U], firstv[JOIN_U], cntv[JOIN_U], hvv[JOIN_U];
            bool mine[JOIN_U];
#pragma unroll
            for (int u = 0; u < JOIN_U; u++) {
                const uint32_t s = s0 + u * JOIN_THREADS;
                kqv[u] = (FULL || s < nq) ? pk[s] : 0u;
            }
#pragma unroll
            for (int u = 0; u < JOIN_U; u++) {
                const uint32_t s = s0 + u * JOIN_THREADS;
                const uint32_t mx = kmer_mix(kqv[u] & SK_SEED_MASK);
                const uint32_t b = mx >> bsh;
                remv[u] = mx & rmask;
                mine[u] = (FULL || s < nq) && (WHOLE || (b >= bb0 && b < bb1));      // else: this seed's bucket belongs to another pass
                uint32_t lo = 0u, hi = 0u;
                if (mine[u]) {
                    if (FITS) {
                        // group word: first seed of the group | the four bucket sizes above it
                        const uint32_t w = s_grp[(b - bb0) >> 2], sh = ((b - bb0) & 3u) * 4u, sizes = w >> 16;
                        const uint32_t below = sizes & ((1u << sh) - 1u);
                        const uint32_t ne = (sizes >> sh) & 15u;
                        lo = (w & 0xFFFFu) + (below & 15u) + ((below >> 4) & 15u) + ((below >> 8) & 15u);
                        hi = lo + ne;
                        if (ne == 15u) { lo = rb[b] - base; hi = rb[b + 1] - base; }       // 15 or more, or behind such a bucket in its group: the global table knows
                    } else { lo = rb[b] - base; hi = rb[b + 1] - base; }
                }
                lov[u] = lo; hiv[u] = hi;
            }
            bool any_multi = false;
#pragma unroll
            for (int u = 0; u < JOIN_U; u++) {
                uint32_t cnt = 0, first = 0;
                if (FITS) {
                    // buckets hold 1-2 seeds on average, equal k-mers side by side: the first JOIN_PROBE_N entries are compared without
                    // a loop (reads clamped into the table: s_fp has slack behind the last seed), longer buckets continue.  The loop
                    // costs the whole wavefront its longest lane: with 1.5 seeds per bucket on average 19 % of the lanes have more than
                    // two entries (some lane of 64 practically always, the longest of them 6-7), 2 % more than four
                    const uint32_t lo = lov[u], ne = hiv[u] - lo, rem = remv[u];
                    uint32_t fe[JOIN_PROBE_N];
#pragma unroll
                    for (int k = 0; k < JOIN_PROBE_N; k++) fe[k] = s_fp[lo + k];
                    first = lo + JOIN_PROBE_N - 1;
#pragma unroll
                    for (int k = JOIN_PROBE_N - 1; k >= 0; k--) {
                        const bool mk = ne > (uint32_t)k && fe[k] == rem;
                        cnt += (uint32_t)mk;
                        first = mk ? lo + (uint32_t)k : first;
                    }
                    if (ne > JOIN_PROBE_N) {
                        for (uint32_t e = lo + JOIN_PROBE_N; e < hiv[u]; e++) {
                            if (s_fp[e] == rem) { if (!cnt) first = e; cnt++; }
                        }
                    }
                } else {
                    const uint32_t kmer = kqv[u] & SK_SEED_MASK;
                    for (uint32_t e = lov[u]; e < hiv[u]; e++) {
                        const uint32_t k2 = rk[base + e] & SK_SEED_MASK;
                        if (k2 == kmer) { if (!cnt) first = e; cnt++; }
                        else if (k2 > kmer) break;
                    }
                }
                if (cnt > rrep) cnt = 0;
                cntv[u] = cnt; firstv[u] = first;
                any_multi |= cnt > 1;
            }
            // all position gathers in flight together: unconditional loads (seeds without a single hit read
            // the genome's first entry, one broadcast address), combined only after the last one is issued
            uint32_t gv[JOIN_U];
#pragma unroll
            for (int u = 0; u < JOIN_U; u++) {
                const uint32_t e = cntv[u] == 1 ? firstv[u] : 0u;
                gv[u] = rg[base + e];
            }
#pragma unroll
            for (int u = 0; u < JOIN_U; u++) {
                // bit 31 of a stag entry is the strand of the indexed k-mer: the hit is reversed when the two differ
                hvv[u] = cntv[u] == 1 ? (gv[u] ^ (kqv[u] & USED_BIT)) : (cntv[u] > 4 ? HIT_MANY : HIT_NONE);
            }
            if (any_multi) {
#pragma unroll
                for (int u = 0; u < JOIN_U; u++) {
                    const uint32_t cnt = cntv[u], first = firstv[u];
                    if (cnt < 2 || cnt > 4) continue;
                    const uint32_t slot = atomicAdd(&pair_nmulti[p], 1u);
                    if (slot < pd.multi_cap) {
                        uint32_t v[4] = {HIT_NONE, HIT_NONE, HIT_NONE, HIT_NONE};
                        for (uint32_t w = 0; w < cnt; w++) v[w] = rg[base + first + w] ^ (kqv[u] & USED_BIT);
                        multi[pd.multi_base + slot] = make_uint4(v[0], v[1], v[2], v[3]);
                        hvv[u] = HIT_MULTI | slot;
                    } else {
                        hvv[u] = HIT_MANY;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < JOIN_U; u++)
                if (mine[u]) hit[s0 + u * JOIN_THREADS] = hvv[u];
        };
        const uint32_t per_trip = JOIN_U * JOIN_THREADS, nfull = nq / per_trip * per_trip;
        uint32_t s0 = tid;
        for (; s0 < nfull; s0 += per_trip) trip(std::true_type{}, s0);
        if (s0 < nq) trip(std::false_type{}, s0);
    }
}

// The same pass for 16-bit remainders in LDS (every genome of 16 K seeds and more: the benchmark's), written for the
// instruction count -- the kernel is issue-bound at two workgroups per CU:
//  * a lane takes FOUR CONSECUTIVE seeds of the chunked genome: one 16-byte load of their k-mers and one 16-byte store of
//    their hit words per trip (the hit words of a pair start at an entry congruent to the genome's seed offset mod 4, as
//    run_extract_kernel relies on too), instead of four 4-byte accesses with a 64-bit address each;
//  * the start of the bucket = the group's base + the sizes of the buckets below it in the group: ONE v_dot8_u32_u4 over
//    the masked size nibbles;
//  * EIGHT entries are compared without a loop, two halfwords per instruction: xor with the remainder in both halves,
//    v_pk_min_u16 against 1 turns every half into "differs", three shift-ors and one shift gather the eight bits, one
//    three-operand bit operation masks them with the bucket size -- count and first match are a population count and a
//    find-first-bit.  (Per entry compares with a loop behind the fourth cost 27 instructions and, because some lane of a
//    wavefront practically always has a fifth entry, a loop trip or two per wavefront: ~ 50 where this is 20.)  Buckets
//    of more than eight entries and buckets marked "look it up in the global table" share one rare loop.
__device__ __forceinline__ uint32_t halves_differ(uint32_t x)     // 1 in bit 0 / bit 16 where the half is not 0
{
    uint32_t r;
    asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]" : "=v"(r) : "v"(x));      // (written out: the compiler turns min(half, 1) into a compare, a select and a permute per half)
    return r;
}

// JOIN_SUB: sub-trips (four consecutive seeds per lane each) between two drains of the memory queue
template <bool WHOLE, int JOIN_SUB>
__device__ __forceinline__ void join_pass16(const SetView &A, const SetView &B, const PairDesc *__restrict__ pairs, const JoinGroup g,
                                            uint32_t *__restrict__ hits, uint4 *__restrict__ multi, uint32_t *__restrict__ pair_nmulti,
                                            const uint16_t *s_fp, const uint32_t *s_grp, const uint32_t *__restrict__ rb,
                                            const uint32_t *__restrict__ rg, uint32_t base, uint32_t bits, uint32_t bb0, uint32_t bb1,
                                            uint32_t rrep, uint32_t tid)
{
    const uint32_t bsh = 30u - bits, rmask = (1u << bsh) - 1u;
    for (uint32_t p = g.pair_begin; p < g.pair_end; p++) {
        const PairDesc pd = pairs[p];
        const SetView &QS = (pd.flags & 2u) ? B : A;
        const GenomeMeta *Qm = QS.meta + pd.q;
        const uint32_t nq = Qm->n_seeds, a = (uint32_t)(Qm->seed_off & 3u), nv = nq + a;
        // virtual seed index v = s + a: v = 0 sits on a 16-byte boundary of both streams
        const uint32_t *pk_al = QS.pkmer + (Qm->seed_off - a);
        uint32_t *hit_al = hits + (pd.hit_base - a);
        const uint32_t last_vec = nv >= 4u ? (nv - 4u) & ~3u : 0u;      // the pair's last whole vector
        // A trip = JOIN_SUB sub-trips with all k-mer loads at its start and all hit-word stores at its end (gfx9 counts loads and
        // stores in one counter and they complete out of order with respect to each other: with a store pending, every wait for a
        // load is a full drain of the queue).  Measured on one box (profiles/round3_join_probe.json): per-entry probe 20.1 ms per
        // step, this probe with ONE sub-trip 18.45, with two 19.5 (a pair of 24 K seeds is three trips of 8 K then, the last one
        // partly idle, and the extra registers spill in the prologue): one is shipped.  The same file holds what the kernel's time is
        // made of: without its gathers 13.9 ms, without its stores 14.4, without both 13.1, without any global memory
        // access 12.4 -- instruction issue + LDS and the 59 GB of HBM traffic (3.0 TB/s) each take about 13 ms and eight wavefronts
        // per SIMD overlap them only partly.
        for (uint32_t v0 = 4u * tid; v0 < nv; v0 += 4u * JOIN_THREADS * JOIN_SUB) {
            uint32_t kq[JOIN_SUB][4], hv[JOIN_SUB][4];
            bool in[JOIN_SUB][4], minev[JOIN_SUB][4], full[JOIN_SUB];
            // unconditional 16-byte loads (no branch between them: they are issued together): a lane at or behind the pair's last
            // whole vector re-reads that one; the one lane with the partial vector at the end reloads its seeds one by one
#pragma unroll
            for (int j = 0; j < JOIN_SUB; j++) {
                const uint32_t vj = v0 + (uint32_t)j * 4u * JOIN_THREADS;
                uint4 k4 = make_uint4(0u, 0u, 0u, 0u);
                if (nv >= 4u) k4 = *reinterpret_cast<const uint4 *>(pk_al + (vj < last_vec ? vj : last_vec));      // (wave-uniform: a pair of fewer than four entries has no whole vector)
                kq[j][0] = k4.x; kq[j][1] = k4.y; kq[j][2] = k4.z; kq[j][3] = k4.w;
            }
#pragma unroll
            for (int j = 0; j < JOIN_SUB; j++) {
                const uint32_t vj = v0 + (uint32_t)j * 4u * JOIN_THREADS;
                full[j] = vj >= a && vj + 4u <= nv;
#pragma unroll
                for (int u = 0; u < 4; u++) in[j][u] = vj + u >= a && vj + u < nv;
                if (vj < nv && vj + 4u > nv) {
#pragma unroll
                    for (int u = 0; u < 4; u++) kq[j][u] = in[j][u] ? pk_al[vj + u] : 0u;
                }
            }
#pragma unroll
            for (int j = 0; j < JOIN_SUB; j++) {
                // in phases, so that the LDS reads of the four seeds are in flight together: group words, bucket entries, compares
                bool any_multi = false;
                uint32_t wv[4], remv[4], lov[4], nev[4], cntv[4], firstv[4];
                uint2 d0v[4], d1v[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t mx = kmer_mix(kq[j][u] & SK_SEED_MASK);
                    const uint32_t b = mx >> bsh;
                    remv[u] = mx & rmask;
                    minev[j][u] = in[j][u] && (WHOLE || (b >= bb0 && b < bb1));      // else: this seed's bucket belongs to another pass
                    const uint32_t bi = WHOLE ? b : (minev[j][u] ? b - bb0 : 0u);
                    wv[u] = s_grp[bi >> 2];                                           // group word: first seed of the group | the four bucket sizes above it
                    lov[u] = (bi & 3u) * 4u;
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t w = wv[u], sh = lov[u], sizes = w >> 16;
                    const uint32_t ne = (sizes >> sh) & 15u;
                    const uint32_t lo = __builtin_amdgcn_udot8(sizes & ((1u << sh) - 1u), 0x11111111u, w & 0xFFFFu, false);
                    nev[u] = ne; lov[u] = lo;
                    // (the second half only where the bucket has it: 2 % of the lanes)
                    d1v[u] = make_uint2(0u, 0u);
                    __builtin_memcpy(&d0v[u], s_fp + lo, 8);
                    if (ne > 4u) __builtin_memcpy(&d1v[u], s_fp + lo + 4, 8);
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t rem = remv[u], ne = nev[u], lo = lov[u];
                    const uint32_t rem2 = rem | (rem << 16);
                    uint32_t t = halves_differ(d0v[u].x ^ rem2);
                    t |= halves_differ(d0v[u].y ^ rem2) << 2;
                    t |= halves_differ(d1v[u].x ^ rem2) << 4;
                    t |= halves_differ(d1v[u].y ^ rem2) << 6;                     // entry e differs: bit e (even e), bit 15 + e (odd e)
                    const uint32_t match = ~(t | (t >> 15)) & ((1u << ne) - 1u) & 0xFFu;
                    uint32_t cnt = (uint32_t)__popc(match), first = lo + (uint32_t)__builtin_ctz(match | 0x100u);
                    if (ne > 8u) {
                        // a long bucket, or one whose bounds only the global table knows (15 seeds or more, or behind such a bucket)
                        uint32_t e = lo + 8u, hi = lo + ne;
                        if (ne == 15u) {
                            const uint32_t b = kmer_mix(kq[j][u] & SK_SEED_MASK) >> bsh;
                            e = rb[b] - base; hi = rb[b + 1] - base; cnt = 0;
                        }
                        for (; e < hi; e++)
                            if (s_fp[e] == rem) { if (!cnt) first = e; cnt++; }
                    }
                    if (!minev[j][u] || cnt > rrep) cnt = 0;
                    cntv[u] = cnt; firstv[u] = first;
                    any_multi |= cnt > 1;
                }
                // all position gathers in flight together: unconditional loads (seeds without a single hit read
                // the genome's first entry, one broadcast address), combined only after the last one is issued
                uint32_t gv[4];
#pragma unroll
                for (int u = 0; u < 4; u++) gv[u] = rg[base + (cntv[u] == 1 ? firstv[u] : 0u)];
#pragma unroll
                for (int u = 0; u < 4; u++)      // bit 31 of a stag entry is the strand of the indexed k-mer: the hit is reversed when the two differ
                    hv[j][u] = cntv[u] == 1 ? (gv[u] ^ (kq[j][u] & USED_BIT)) : (cntv[u] > 4 ? HIT_MANY : HIT_NONE);
                if (any_multi) {
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const uint32_t cnt = cntv[u], first = firstv[u];
                        if (cnt < 2 || cnt > 4) continue;
                        const uint32_t slot = atomicAdd(&pair_nmulti[p], 1u);
                        if (slot < pd.multi_cap) {
                            uint32_t v[4] = {HIT_NONE, HIT_NONE, HIT_NONE, HIT_NONE};
                            for (uint32_t w = 0; w < cnt; w++) v[w] = rg[base + first + w] ^ (kq[j][u] & USED_BIT);
                            multi[pd.multi_base + slot] = make_uint4(v[0], v[1], v[2], v[3]);
                            hv[j][u] = HIT_MULTI | slot;
                        } else {
                            hv[j][u] = HIT_MANY;
                        }
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < JOIN_SUB; j++) {
                const uint32_t vj = v0 + (uint32_t)j * 4u * JOIN_THREADS;
                if (WHOLE && full[j]) {
                    *reinterpret_cast<uint4 *>(hit_al + vj) = make_uint4(hv[j][0], hv[j][1], hv[j][2], hv[j][3]);
                } else {
#pragma unroll
                    for (int u = 0; u < 4; u++)
                        if (minev[j][u]) hit_al[vj + u] = hv[j][u];
                }
            }
        }
    }
}

// one bucket range of R after the other: stage, probe
template <typename FP, int V2>
__device__ __forceinline__ void join_group(const SetView &A, const SetView &B, const PairDesc *__restrict__ pairs, const JoinGroup g,
                                           uint32_t *__restrict__ hits, uint4 *__restrict__ multi, uint32_t *__restrict__ pair_nmulti,
                                           unsigned char *smem, uint32_t smem_bytes, const SetView &RS, const GenomeMeta *Rm, uint32_t tid)
{
    const uint32_t bits = Rm->bucket_bits, nbk = 1u << bits, rrep = Rm->rep_cut;
    const uint32_t *rk = RS.skmer + Rm->seed_off, *rg = RS.stag + Rm->seed_off, *rb = RS.boff + Rm->bucket_off;   // rg: position | record tag | strand
    // LDS: [control words | group words (one per four buckets: first seed of the group relative to the pass, four sizes) | remainders]
    uint32_t *s_ctl = reinterpret_cast<uint32_t *>(smem);           // [0] = end bucket of the pass
    uint32_t *s_grp = reinterpret_cast<uint32_t *>(smem + 64);
    // whole table in one pass if it fits; else as many buckets as half of the space takes, at most 65535 seeds per pass
    const uint32_t whole_off = nbk + 64u;
    const bool one = whole_off + ((size_t)Rm->n_seeds + JOIN_SLACK) * sizeof(FP) <= smem_bytes;
    const uint32_t bcap = one ? nbk : ((smem_bytes / 2u - 64u) & ~3u);                   // buckets held per pass (a multiple of 4)
    const uint32_t fp_off = one ? whole_off : smem_bytes / 2u;
    FP *s_fp = reinterpret_cast<FP *>(smem + fp_off);
    uint32_t kcap = (smem_bytes - fp_off) / (uint32_t)sizeof(FP) - JOIN_SLACK;                   // remainders held per pass
    kcap = kcap < 65535u ? kcap : 65535u;
    const uint32_t bsh = 30u - bits, rmask = (1u << bsh) - 1u;

    for (uint32_t bb0 = 0; bb0 < nbk;) {
        __syncthreads();
        if (tid == 0) {
            // the largest bucket range [bb0, bb1), whole groups of four, whose remainders fit; at least one group
            uint32_t hi = bb0 + bcap < nbk ? bb0 + bcap : nbk, lo = bb0 + 4u;
            const uint32_t base = rb[bb0];
            if (rb[hi] - base > kcap) {
                while (lo < hi) {   // largest bb1 in [bb0+4, hi], a multiple of 4, with rb[bb1] - base <= kcap
                    const uint32_t mid = ((lo + hi) / 2u + 3u) & ~3u;
                    if (rb[mid] - base <= kcap) lo = mid; else hi = mid - 4u;
                }
                hi = lo;
            }
            s_ctl[0] = hi;
        }
        __syncthreads();
        const uint32_t bb1 = s_ctl[0];
        const uint32_t base = rb[bb0], nk = rb[bb1] - base;
        const bool fits = nk <= kcap;   // false only for one group of four buckets with more than kcap seeds
        if (fits) {
            for (uint32_t i = tid; i < (bb1 - bb0) / 4u; i += JOIN_THREADS) {
                const uint32_t b = bb0 + 4u * i;
                const uint32_t o0 = rb[b], o1 = rb[b + 1], o2 = rb[b + 2], o3 = rb[b + 3], o4 = rb[b + 4];
                const uint32_t c0 = o1 - o0, c1 = o2 - o1, c2 = o3 - o2, c3 = o4 - o3;
                // a bucket of 15 seeds or more is marked 15 and looked up in the global table; the sizes in the word no longer add up
                // to the starts of the buckets BEHIND it in the group, so those are marked 15 as well
                const bool v0 = c0 >= 15u, v1 = v0 || c1 >= 15u, v2 = v1 || c2 >= 15u, v3 = v2 || c3 >= 15u;
                s_grp[i] = (o0 - base) | ((v0 ? 15u : c0) << 16) | ((v1 ? 15u : c1) << 20) | ((v2 ? 15u : c2) << 24) | ((v3 ? 15u : c3) << 28);
            }
            for (uint32_t i = tid; i < nk; i += JOIN_THREADS) s_fp[i] = (FP)(kmer_mix(rk[base + i] & SK_SEED_MASK) & rmask);
            if (tid < JOIN_SLACK) s_fp[nk + tid] = (FP)0;
        }
        __syncthreads();
        const bool whole = bb0 == 0 && bb1 == nbk;
        if (V2 != 0 && sizeof(FP) == 2 && fits) {
            if (whole) join_pass16<true, (V2 > 1 ? V2 : 1)>(A, B, pairs, g, hits, multi, pair_nmulti, reinterpret_cast<const uint16_t *>(s_fp), s_grp, rb, rg, base, bits, bb0, bb1, rrep, tid);
            else join_pass16<false, 1>(A, B, pairs, g, hits, multi, pair_nmulti, reinterpret_cast<const uint16_t *>(s_fp), s_grp, rb, rg, base, bits, bb0, bb1, rrep, tid);
        }
        else if (fits && whole) join_pass<FP, true, true>(A, B, pairs, g, hits, multi, pair_nmulti, s_fp, s_grp, rb, rk, rg, base, bits, bb0, bb1, rrep, tid);
        else if (fits) join_pass<FP, true, false>(A, B, pairs, g, hits, multi, pair_nmulti, s_fp, s_grp, rb, rk, rg, base, bits, bb0, bb1, rrep, tid);
        else join_pass<FP, false, false>(A, B, pairs, g, hits, multi, pair_nmulti, s_fp, s_grp, rb, rk, rg, base, bits, bb0, bb1, rrep, tid);
        bb0 = bb1;
    }
}

// (amdgpu_waves_per_eu(8): TWO of these 1024-thread workgroups per CU need 8 wavefronts per SIMD, i.e. at most 64 VGPRs and -- the
// limit that was silently missed before -- at most 80 SGPRs per wavefront.  With the two SetViews in scalar registers the compiler
// took 101, which admits 6 wavefronts per SIMD: ONE workgroup per CU, half the wavefronts this latency-bound kernel was designed for.)
// V2: 0 = round 2's per-entry probe (SKDER_AMD_JOIN_V1), else the sub-trips per trip of join_pass16
template <int V2>
__global__ __launch_bounds__(JOIN_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void join_probe_kernel(SetView A, SetView B, const PairDesc *__restrict__ pairs,
                                                                  const JoinGroup *__restrict__ groups,
                                                                  uint32_t *__restrict__ hits, uint4 *__restrict__ multi,
                                                                  uint32_t *__restrict__ pair_nmulti, uint32_t smem_bytes)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char join_smem[];
    const uint32_t tid = threadIdx.x;
    const JoinGroup g = groups[blockIdx.x];
    const PairDesc pd0 = pairs[g.pair_begin];
    const SetView &RS = (pd0.flags & 4u) ? B : A;
    const GenomeMeta *Rm = RS.meta + pd0.r;
    // 16-bit remainders identify a k-mer inside its bucket once there are 2^14 buckets (30 - bits <= 16)
    if (Rm->bucket_bits >= 14u) join_group<uint16_t, V2>(A, B, pairs, g, hits, multi, pair_nmulti, join_smem, smem_bytes, RS, Rm, tid);
    else join_group<uint32_t, V2>(A, B, pairs, g, hits, multi, pair_nmulti, join_smem, smem_bytes, RS, Rm, tid);
}

// ---------------------------------------------------------------------------------------------
// RUN EXTRACTION: seed-parallel, coalesced.  Most classified seeds of related genomes merely continue the
// previous hit -- same record and strand, 1..2500 bases on, ahead on the other genome, and on the same
// diagonal or at most RUN_GAP bases off it (real genomes carry a short indel every few hundred bases) --
// so the chaining kernels are fed RUNS, maximal stretches of such seeds, instead of one word per seed.
// Why RUN_GAP = 10 = anchor score / 2: along a run every link scores 20 - gap >= 10, and the offer of any
// other anchor (constant score + 20 - diagonal distance) moves by at most the link's gap <= 10, so (1) inside
// a run the previous anchor is always the best predecessor of the next one (nearest on ties) and (2) a
// competitor that cannot beat the run at its second anchor never can (chain_runs_kernel checks that once).
// One workgroup per pair, its four wavefronts independent of one another: wave w takes the w-th quarter of
// the pair's seeds, a SEGMENT of 256 consecutive seeds at a time (4 per lane: one 16-byte load per input
// stream and lane), and classifies every seed against the previous hit (inside the lane in registers,
// across lanes by ballots and shuffles, across segments in wave-uniform registers).  A record is written
// where a run STARTS and carries, besides its first seed, the hit in front of it and the wave's running
// counts of hits and diagonal steps up to there: the end, the length and the step sum of a run are read
// off the NEXT record, so no reduction over a run is needed.  Wave w writes into the w-th quarter of the
// pair's record region and closes it with a LINK record (next: the following quarter) or, the last one, a
// TERMINATOR; both close the run in front of them.  A run never crosses a chunk boundary or a quarter; the
// chaining kernel joins such pieces again through its ordinary look-back.  A seed with 2..4 occurrences, or
// too many, is a record of its own.  The first record of every chunk is registered in chunk_rec0.  A quarter
// with more records than it holds is marked: the chunks with seeds in it take the slow path.
struct __attribute__((aligned(16))) RunRec {
    uint32_t qi, q0, hw, cn;      // first seed: index in the chunked genome, position, hit word (or HIT_MULTI | slot, HIT_MANY); hits of the quarter in front of it
    uint32_t pq, pw, pqi, cg;     // the hit in front of it: position, hit word, seed index; diagonal steps of the quarter in front of it
};
#define SEG_SEEDS 256u
#define RUN_GAP 10
#define REC_LINK 0xFFFFFFFEu      // qi of a link record; its q0 is the index of the next record, its hw the first seed of the next quarter
#define REC_END 0xFFFFFFFFu       // qi of the terminator
static_assert(2 * RUN_GAP <= ANI_ANCHOR_SCORE, "run links must keep at least half of the anchor score");

__global__ __launch_bounds__(256) void run_extract_kernel(SetView A, SetView B, const PairDesc *__restrict__ pairs,
                                                          const uint32_t *__restrict__ hits, RunRec *__restrict__ recs,
                                                          uint32_t *__restrict__ pair_over, uint32_t *__restrict__ chunk_rec0)
{
    const uint32_t pid = blockIdx.x;       // (an order by chunked genome, per XCD, to share its positions in L2 measured no faster)
    const PairDesc pd = pairs[pid];
    const SetView &QS = (pd.flags & 2u) ? B : A;
    const GenomeMeta *Qm = QS.meta + pd.q;
    const uint32_t nq = Qm->n_seeds, a = (uint32_t)(Qm->seed_off & 3u), nv = nq + a;
    // virtual seed index v = s + a: v = 0 sits on a 16-byte boundary of all three streams (the hit words of a pair
    // start at an entry congruent to the genome's seed offset)
    const uint32_t *qg_al = QS.pgpos + (Qm->seed_off - a);
    const uint8_t *cs_al = QS.pcs + (Qm->seed_off - a);
    const uint32_t *ck_of = QS.pchunk + Qm->seed_off;
    const uint32_t *hit_al = hits + (pd.hit_base - a);
    uint32_t *rec0 = chunk_rec0 + pd.chunk_base;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t nseg = (nv + SEG_SEEDS - 1u) / SEG_SEEDS, per = (nseg + 3u) / 4u;
    const uint32_t sg_lo = wv * per < nseg ? wv * per : nseg, sg_hi = (wv + 1u) * per < nseg ? (wv + 1u) * per : nseg;
    const uint32_t cap4 = pd.rec_cap / 4u, reg0 = wv * cap4;           // this wave's quarter of the pair's record region
    RunRec *out_base = recs + pd.rec_base;
    uint32_t run_rec = 0, run_nm = 0, run_g = 0;                     // totals of the segments so far (wave-uniform)
    uint32_t car_q = 0, car_w = HIT_NONE, car_v = 0xFFFFFFFFu;       // last hit of the segments so far
    bool car_ok = false;                                             // there is such a hit and no chunk began since
    bool overflow = false;
    const unsigned long long lowbits = (1ull << lane) - 1ull;
    for (uint32_t sg = sg_lo; sg < sg_hi; sg++) {
        const uint32_t v0 = sg * SEG_SEEDS + lane * 4u;
        uint32_t hv[4], qv[4], csw = 0;
        if (v0 >= a && v0 + 4u <= nv) {
            const uint4 h4 = *reinterpret_cast<const uint4 *>(hit_al + v0);
            const uint4 q4 = *reinterpret_cast<const uint4 *>(qg_al + v0);
            csw = *reinterpret_cast<const uint32_t *>(cs_al + v0);
            hv[0] = h4.x; hv[1] = h4.y; hv[2] = h4.z; hv[3] = h4.w;
            qv[0] = q4.x; qv[1] = q4.y; qv[2] = q4.z; qv[3] = q4.w;
        } else {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t v = v0 + u;
                const bool in = v >= a && v < nv;
                hv[u] = in ? hit_al[v] : HIT_NONE;
                qv[u] = in ? qg_al[v] : 0u;
                csw |= in ? (uint32_t)cs_al[v] << (8 * u) : 0u;
            }
        }
        // A. the lane's own four seeds: hit or not, chunk start or not, and the DIAGONAL WORD of a hit -- the hit word with
        // the position replaced by one value per diagonal (position - q forward, -position - 1 - q reverse, modulo
        // 2^32 across the record tag and strand above it): for two hits of the same record and strand the difference
        // of the words is the difference of their diagonals, and q - q' plus that difference is how far the second
        // lies AHEAD of the first on the other genome (in the direction of the strand)
        bool nm[4], cs[4];
        uint32_t yv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            nm[u] = hv[u] != HIT_NONE;
            cs[u] = ((csw >> (8 * u)) & 1u) != 0u;
            const uint32_t sgn = (uint32_t)((int32_t)hv[u] >> 31);
            yv[u] = (hv[u] ^ (sgn & HIT_POS_MASK)) - qv[u];
        }
        const bool has_nm = nm[0] | nm[1] | nm[2] | nm[3];
        const uint32_t ul = nm[3] ? 3u : (nm[2] ? 2u : (nm[1] ? 1u : 0u));                       // the lane's last hit
        const uint32_t w_l = nm[3] ? hv[3] : (nm[2] ? hv[2] : (nm[1] ? hv[1] : hv[0]));
        const uint32_t q_l = nm[3] ? qv[3] : (nm[2] ? qv[2] : (nm[1] ? qv[1] : qv[0]));
        // a chunk starts behind the lane's last hit (anywhere, if the lane has none): the next hit cannot continue
        bool tail_cs = false;
#pragma unroll
        for (int u = 0; u < 4; u++) tail_cs = (tail_cs | cs[u]) & !nm[u];
        const unsigned long long M = __ballot(has_nm), T = __ballot(tail_cs);
        // B. the previous hit: from the nearest lane below that has one, else the last hit of the segments before
        const unsigned long long below = M & lowbits, tbelow = T & lowbits;
        const bool pin = below != 0ull;
        const uint32_t P = pin ? 63u - (uint32_t)__clzll((long long)below) : 0u;
        const uint32_t sw = (uint32_t)__shfl((int)w_l, (int)P, 64), sq = (uint32_t)__shfl((int)q_l, (int)P, 64);
        const uint32_t sv = (uint32_t)__shfl((int)(v0 + ul), (int)P, 64);
        const uint32_t pw_in = pin ? sw : car_w, pq_in = pin ? sq : car_q, pv_in = pin ? sv : car_v;
        // no chunk start between that hit and this lane: no tail flag in the lanes [P, lane)
        const bool pv = pin ? tbelow < (1ull << P) : (car_ok && tbelow == 0ull);
        // C. run starts among the lane's seeds; diagonal step of every continuing seed.  first: starts that may be the
        // first record of their chunk (a chunk began since the previous hit, or there is none).  Straight-line code:
        // every seed is classified, the results of the missing ones are masked out
        bool st[4], fi[4];
        uint32_t gl[4];
        {
            uint32_t pw = pw_in, pq = pq_in;
            uint32_t py = (pw ^ ((uint32_t)((int32_t)pw >> 31) & HIT_POS_MASK)) - pq;
            bool pending = !pv;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                pending = pending | cs[u];
                const uint32_t w = hv[u], q = qv[u], y = yv[u];
                const int32_t dd = (int32_t)(y - py);                       // diagonal step (same record and strand)
                const uint32_t dq = q - pq;
                const uint32_t kb = (w ^ pw) | (w & 0x40000000u);           // < 2^24: same record and strand, both single hits
                const bool cont = nm[u] & !pending & (kb < (1u << HIT_POS_BITS)) & ((uint32_t)(dd + RUN_GAP) <= 2u * RUN_GAP) &
                                  (dq <= (uint32_t)ANI_BP_BAND) & ((int32_t)(dq + (uint32_t)dd) > 0);
                st[u] = nm[u] & !cont;
                fi[u] = nm[u] & pending;
                const int32_t ad = dd < 0 ? -dd : dd;
                gl[u] = cont ? (uint32_t)ad : 0u;
                pw = nm[u] ? w : pw; pq = nm[u] ? q : pq; py = nm[u] ? y : py;
                pending = pending & !nm[u];
            }
        }
        // D. running counts in front of the lane: hits, records, diagonal steps (one packed scan)
        const uint32_t cnt_l = (uint32_t)nm[0] + (uint32_t)nm[1] + (uint32_t)nm[2] + (uint32_t)nm[3];
        const uint32_t nrec_l = (uint32_t)st[0] + (uint32_t)st[1] + (uint32_t)st[2] + (uint32_t)st[3];
        const uint32_t g_l = gl[0] + gl[1] + gl[2] + gl[3];
        uint32_t tot;
        const uint32_t ex = wave_excl_scan(cnt_l | (nrec_l << 10) | (g_l << 20), tot);
        const uint32_t seg_rec = (tot >> 10) & 1023u;
        if (run_rec + seg_rec + 1u > cap4) { overflow = true; break; }       // + the closing record; wave-uniform
        if (nrec_l) {
#define SEL4(X, I) ((I) == 0 ? X[0] : ((I) == 1 ? X[1] : ((I) == 2 ? X[2] : X[3])))
            const uint32_t nmmask = (uint32_t)nm[0] | ((uint32_t)nm[1] << 1) | ((uint32_t)nm[2] << 2) | ((uint32_t)nm[3] << 3);
            const uint32_t startmask = (uint32_t)st[0] | ((uint32_t)st[1] << 1) | ((uint32_t)st[2] << 2) | ((uint32_t)st[3] << 3);
            const uint32_t firstmask = (uint32_t)fi[0] | ((uint32_t)fi[1] << 1) | ((uint32_t)fi[2] << 2) | ((uint32_t)fi[3] << 3);
            const uint32_t pex = ex & 1023u, rex = (ex >> 10) & 1023u, gex = ex >> 20;
            uint32_t sm = startmask, j = 0;
            while (sm) {
                const uint32_t u = (uint32_t)__ffs((int)sm) - 1u;
                sm &= sm - 1u;
                RunRec r;
                r.qi = v0 + u - a; r.q0 = SEL4(qv, u); r.hw = SEL4(hv, u);
                const uint32_t lowm = nmmask & ((1u << u) - 1u);          // the lane's hits in front of this one
                r.cn = run_nm + pex + (uint32_t)__popc(lowm);
                uint32_t gs = 0;
#pragma unroll
                for (int x = 0; x < 4; x++) gs += ((uint32_t)x < u) ? gl[x] : 0u;
                r.cg = run_g + gex + gs;
                if (lowm) {
                    const uint32_t lu = 31u - (uint32_t)__clz((int)lowm);
                    r.pq = SEL4(qv, lu); r.pw = SEL4(hv, lu); r.pqi = v0 + lu - a;
                } else { r.pq = pq_in; r.pw = pw_in; r.pqi = pv_in == 0xFFFFFFFFu ? 0xFFFFFFFFu : pv_in - a; }
                const uint32_t at = reg0 + run_rec + rex + j;
                out_base[at] = r;
                if ((firstmask >> u) & 1u) atomicMin(&rec0[ck_of[r.qi]], at);      // (a plain store where no other quarter can have the chunk: measured slower, 11.9 against 10.8 ms)
                j++;
            }
#undef SEL4
        }
        // E. carry into the next segment
        run_nm += tot & 1023u; run_rec += seg_rec; run_g += tot >> 20;
        if (M) {
            const uint32_t topl = 63u - (uint32_t)__clzll((long long)M);
            car_q = (uint32_t)__shfl((int)q_l, (int)topl, 64); car_w = (uint32_t)__shfl((int)w_l, (int)topl, 64);
            car_v = (uint32_t)__shfl((int)(v0 + ul), (int)topl, 64);
            car_ok = (T >> topl) == 0ull;
        } else {
            car_ok = car_ok && T == 0ull;
        }
    }
    if (overflow) { if (lane == 0) atomicOr(&pair_over[pid], 1u << wv); return; }      // the chunks of this quarter take the slow path
    if (lane == 0) {
        RunRec r;       // closes the last run of the quarter; leads on to the next quarter, or ends the pair
        // a link also says where the next quarter's seeds begin: a chunk that ends in front of them is finished at the link,
        // and nobody has to look into a quarter that may have overflowed (its region then holds stale records)
        const uint32_t next_v = sg_hi * SEG_SEEDS;
        r.qi = wv == 3u ? REC_END : REC_LINK; r.q0 = (wv + 1u) * cap4; r.hw = next_v > a ? next_v - a : 0u; r.cn = run_nm;
        r.pq = car_q; r.pw = car_w; r.pqi = car_v == 0xFFFFFFFFu ? 0xFFFFFFFFu : car_v - a; r.cg = run_g;
        out_base[reg0 + run_rec] = r;
    }
}

// ---------------------------------------------------------------------------------------------
// FAST PATH, first sieve: one lane per (pair, 20 kb chunk), a short loop over the chunk's first records.
// A chunk without a hit has no chain.  A chunk whose hits form ONE run of single-occurrence seeds IS its
// chain: inside a run every anchor chains to the one before (run_extract_kernel, "Why RUN_GAP"), scores rise
// along it, so the best end is the last anchor and the back-track takes all n of them -- score 20 n - steps,
// a chain if n >= 3.  With k = 15 about one seed in 180 also hits an unrelated place of the other genome, so
// the usual chunk is such a run cut into pieces by one or two STRAY hits; that is settled here as well:
//   * a PATH: records that follow one another like the seeds of a run do (same record and strand,
//     1..2500 bases on, ahead on the other genome, at most RUN_GAP off the diagonal of the hit in front) --
//     all links cost <= 10, so each anchor chains to the main anchor before it, strays in between or not
//     (they lie at most two anchors deep in the 50-anchor look-back); a chunk may hold up to three paths one
//     after the other (the other genome's records end inside it, or a stretch without hits is longer than
//     the 2500-base band), each of a record / strand of its own or out of reach of the others' anchors, so
//     that nothing chains from one to the next;
//   * at most TWO strays -- a seed that hits an unrelated place only, or the second occurrence of a seed whose
//     other occurrence lies on the main path --, each of another record or strand than the main path or
//     further from its diagonal than max_gap plus all the path's diagonal steps: they can neither give to nor
//     take from a main anchor, and two anchors alone are not a chain (min_anchors = 3).
// Everything else goes on a list for chain_runs_kernel (lanes packed with chunks that need its loop); pairs
// whose records overflowed and pairs that need the unabridged algorithm go to the slow path.
#define SIEVE_RECORDS 6
#define GEN_LISTS 256u
#ifdef SKDER_SIEVE_STATS
#define SIEVE_WHY(I) atomicAdd(counters + 16 + (I), 1u)
#else
#define SIEVE_WHY(I)
#endif
__global__ __launch_bounds__(256) void chain_single_kernel(SetView A, SetView B, const PairDesc *__restrict__ pairs, uint32_t npairs,
                                                           uint32_t total_chunks, const RunRec *__restrict__ recs,
                                                           const uint32_t *__restrict__ pair_over, const uint32_t *__restrict__ chunk_rec0,
                                                           const uint32_t *__restrict__ wg_pair, const uint4 *__restrict__ multi,
                                                           ChainRec *__restrict__ fast_chains,
                                                           uint32_t *__restrict__ chunk_state, uint32_t *__restrict__ slow_list,
                                                           uint32_t *__restrict__ counters, uint32_t *__restrict__ gen_list,
                                                           uint32_t *__restrict__ gen_cnt, uint32_t gen_cap,
                                                           uint32_t *__restrict__ pair_na, int xcd_remap, uint32_t *__restrict__ chunk_pair)
{
    // workgroups in launch order (dealt round-robin to the 8 XCDs): every record is read once, there is nothing an XCD's L2
    // could share, and one contiguous stream over the chip measured 1.8 ms per step faster than an eighth of the list per XCD
    const uint32_t wg = blockIdx.x;
    const uint32_t t = wg * 256u + threadIdx.x;
    const bool in = t < total_chunks;
    uint32_t pi = 0, n_add = 0;
    bool to_gen = false;
    if (in) {
        const uint32_t idx0 = chunk_rec0[t];                 // independent of the descriptor: in flight beside it
        pi = wg_pair[wg];
        // a workgroup's 256 chunks rarely span more than three pairs.  The descriptors of the first two are requested whole
        // at once (the same two addresses for the whole wavefront) and the third one's first chunk beside them: the
        // usual chunk then has its descriptor without a dependent load; only a chunk of the third pair or beyond looks again
        const uint32_t p1 = pi + 1u < npairs ? pi + 1u : pi, p2 = pi + 2u < npairs ? pi + 2u : p1;
        const PairDesc pd0 = pairs[pi], pd1 = pairs[p1];
        const uint32_t cb2 = pairs[p2].chunk_base;
        PairDesc pd = pd0;
        if (p2 != p1 && cb2 <= t) {
            pi = p2;
            while (pi + 1u < npairs && pairs[pi + 1u].chunk_base <= t) pi++;
            pd = pairs[pi];
        } else if (p1 != pi && pd1.chunk_base <= t) { pi = p1; pd = pd1; }
        uint32_t over = pair_over[pi];
        const uint32_t c = t - pd.chunk_base;
        const SetView &QS = (pd.flags & 2u) ? B : A;
        const uint32_t s0 = QS.chunk_start[pd.q_chunk_off + c], s1 = QS.chunk_start[pd.q_chunk_off + c + 1];
        if (over) {      // quarters of the record region that overflowed: only the chunks with seeds in one of them are lost
            const uint32_t qa = ((s0 + pd.seg_a) >> 8) / pd.seg_per, qb = ((s1 - 1u + pd.seg_a) >> 8) / pd.seg_per;
            over &= (2u << qb) - (1u << qa);
        }
        if ((pd.flags & 8u) || over || (xcd_remap & 2)) {
            chunk_state[t] = CHUNK_SLOW;
            chunk_pair[t] = pi;                  // the kernels further down find the chunk's pair without a search
            slow_list[atomicAdd(counters, 1u)] = t;
            atomicAdd(counters + 1 + ((pd.flags & 8u) || (xcd_remap & 2) ? 6 : 8), 1u);
        } else if (idx0 == 0xFFFFFFFFu) {
            chunk_state[t] = 0u;
        } else {
            const uint4 *rp = reinterpret_cast<const uint4 *>(recs + pd.rec_base) + 2u * idx0;
            uint4 a0 = rp[0], a1 = rp[1];
            // the next three records are requested at once (a chunk seldom has more; the region has room behind its last record)
            uint4 f0 = rp[2], f1 = rp[3], f2 = rp[4], f3 = rp[5], f4 = rp[6], f5 = rp[7];
            bool fail = (xcd_remap & 1024) != 0, main_on = false;      // 1024: SKDER_AMD_NO_SIEVE
            uint32_t n = 0, G = 0, nstray = 0, anchors = 0, nfin = 0, npath = 0;
            uint32_t m_qi = 0, m_q0 = 0, m_hw = 0, l_q = 0, l_hw = 0, l_qi = 0;     // current path: first anchor; last anchor
            uint32_t st_hw0 = 0, st_q0 = 0, st_hw1 = 0, st_q1 = 0;                  // the strays
            uint32_t p_hw[FAST_SLOTS] = {0, 0, 0}, p_G[FAST_SLOTS] = {0, 0, 0};     // closed and current paths: key, first diagonal, steps
            int32_t p_D[FAST_SLOTS] = {0, 0, 0};
            uint32_t p_lq[FAST_SLOTS] = {0, 0, 0};                                   // ... and the position of their last anchor
            ChainRec *slots = fast_chains + (uint64_t)t * FAST_SLOTS;
            // the current path ends: its chain, and what the strays have to be checked against
#define CLOSE_PATH()                                                                                                  \
            do {                                                                                                      \
                p_hw[npath] = m_hw; p_G[npath] = G; p_lq[npath] = l_q;                                                \
                p_D[npath] = (m_hw >> 31) ? (int32_t)(m_hw & HIT_POS_MASK) + (int32_t)m_q0 : (int32_t)(m_hw & HIT_POS_MASK) - (int32_t)m_q0; \
                npath++;                                                                                              \
                if (n >= ANI_MIN_ANCHORS) {                                                                           \
                    ChainRec cr;                                                                                      \
                    cr.score = ANI_ANCHOR_SCORE * (int32_t)n - (int32_t)G; cr.n = n; cr.n_seeds = l_qi - m_qi + 1u;   \
                    cr.q0 = m_q0; cr.q1 = l_q;                                                                        \
                    const uint32_t ra = m_hw & HIT_POS_MASK, rb = l_hw & HIT_POS_MASK;                                \
                    cr.r0 = ra < rb ? ra : rb; cr.r1 = ra > rb ? ra : rb;                                             \
                    cr.chunk = c;                                                                                     \
                    slots[nfin++] = cr;                                                                               \
                }                                                                                                     \
            } while (0)
            for (int k = 0; k < SIEVE_RECORDS + 1 && !fail; k++) {
                if (a0.x >= s1) { if (a0.x == REC_LINK && s1 > a0.z) { fail = true; SIEVE_WHY(8); } break; }      // (a link: the chunk may go on in the next quarter)
                if (k == SIEVE_RECORDS || a0.z == HIT_MANY) { fail = true; SIEVE_WHY(k == SIEVE_RECORDS ? 12 : 13); break; }
                rp += 2;
                uint4 b0, b1;                                // the record behind closes this one
                if (k < 3) { b0 = f0; b1 = f1; f0 = f2; f1 = f3; f2 = f4; f3 = f5; }
                else { b0 = rp[0]; b1 = rp[1]; }
                const uint32_t rn = b0.w - a0.w, rg = b1.w - a1.w;
                // does a hit continue the current path behind its last anchor?
#define JOINS(W, GOUT)                                                                                               \
                [&]() -> bool {                                                                                       \
                    const uint32_t sgw = (uint32_t)((int32_t)(W) >> 31), sgp = (uint32_t)((int32_t)l_hw >> 31);       \
                    const int32_t dd = (int32_t)((((W) & HIT_POS_MASK) ^ sgw) - a0.y) - (int32_t)(((l_hw & HIT_POS_MASK) ^ sgp) - l_q); \
                    const int32_t gabs_ = dd < 0 ? -dd : dd;                                                          \
                    const int32_t drs = (int32_t)((W) & HIT_POS_MASK) - (int32_t)(l_hw & HIT_POS_MASK);               \
                    GOUT = (uint32_t)gabs_;                                                                           \
                    return !(((W) ^ l_hw) & HIT_KEY_MASK) && gabs_ <= RUN_GAP && (a0.y - l_q) - 1u < (uint32_t)ANI_BP_BAND && \
                           (sgw ? drs < 0 : drs > 0);                                                                 \
                }()
                if ((a0.z & 0xFF000000u) == HIT_MULTI) {
                    // a seed with two occurrences: usually its place on the current path and a stray
                    const uint4 mv = multi[pd.multi_base + (a0.z & 0x00FFFFFFu)];
                    uint32_t gx = 0, gy = 0;
                    const bool two = mv.z == HIT_NONE && main_on && nstray < 2u;
                    const bool jx = two && JOINS(mv.x, gx), jy = two && JOINS(mv.y, gy);
                    if (jx == jy) { fail = true; SIEVE_WHY(13); break; }
                    const uint32_t wj = jx ? mv.x : mv.y, ws = jx ? mv.y : mv.x;
                    anchors += 2u;
                    n += 1u; G += jx ? gx : gy; l_q = a0.y; l_hw = wj; l_qi = a0.x;
                    if (nstray == 0u) { st_hw0 = ws; st_q0 = a0.y; } else { st_hw1 = ws; st_q1 = a0.y; }
                    nstray++;
                } else {
                    anchors += rn;
                    bool joins = false;
                    uint32_t g = 0;
                    if (main_on) {
                        joins = JOINS(a0.z, g);
                        if (joins) { n += rn; G += rg + g; l_q = b1.x; l_hw = b1.y; l_qi = b1.z; }
                    }
                    if (!joins) {
                        if (rn >= 2u) {
                            // a new path: of a record / strand no path of the chunk had so far, or more than the 2500-base band
                            // behind the last anchor of every path that had it (nothing can chain across)
                            if (main_on) {
                                if (npath + 1u >= FAST_SLOTS) { fail = true; SIEVE_WHY(10); break; }
                                CLOSE_PATH();
                                bool clash = false;
                                for (uint32_t x = 0; x < npath; x++) clash |= !((p_hw[x] ^ a0.z) & HIT_KEY_MASK) && a0.y - p_lq[x] <= (uint32_t)ANI_BP_BAND;
                                if (clash) { fail = true; SIEVE_WHY(10); break; }
                            }
                            main_on = true;
                            m_qi = a0.x; m_q0 = a0.y; m_hw = a0.z; n = rn; G = rg; l_q = b1.x; l_hw = b1.y; l_qi = b1.z;
                        } else if (nstray < 2u) {
                            if (nstray == 0u) { st_hw0 = a0.z; st_q0 = a0.y; } else { st_hw1 = a0.z; st_q1 = a0.y; }
                            nstray++;
                        } else { fail = true; SIEVE_WHY(14); break; }
                    }
                }
#undef JOINS
                a0 = b0; a1 = b1;
            }
            if (!fail && main_on) CLOSE_PATH();
#undef CLOSE_PATH
            if (!fail && nstray) {
                // the strays must be unable to chain with any path
                const int32_t ds0 = (st_hw0 >> 31) ? (int32_t)(st_hw0 & HIT_POS_MASK) + (int32_t)st_q0 : (int32_t)(st_hw0 & HIT_POS_MASK) - (int32_t)st_q0;
                const int32_t ds1 = (st_hw1 >> 31) ? (int32_t)(st_hw1 & HIT_POS_MASK) + (int32_t)st_q1 : (int32_t)(st_hw1 & HIT_POS_MASK) - (int32_t)st_q1;
                for (uint32_t x = 0; x < npath; x++) {
                    const int32_t lim = ANI_MAX_GAP + (int32_t)p_G[x];
                    if (!((st_hw0 ^ p_hw[x]) & HIT_KEY_MASK) && abs(ds0 - p_D[x]) <= lim) fail = true;
                    if (nstray > 1u && !((st_hw1 ^ p_hw[x]) & HIT_KEY_MASK) && abs(ds1 - p_D[x]) <= lim) fail = true;
                }
                if (fail) SIEVE_WHY(15);
            }
            if (!fail) { n_add = anchors; chunk_state[t] = nfin; }
            else { to_gen = true; chunk_pair[t] = pi; }
        }
    }
    {
        // the chunks left for chain_runs_kernel: one atomic per wavefront, the lanes take consecutive places.  Nearly every
        // wavefront has some, and one counter for the whole device would serialise them: GEN_LISTS lists, picked by the
        // workgroup number, each with room for all chunks of the workgroups that use it
        const unsigned long long gm = __ballot(to_gen);
        if (gm) {
            const uint32_t lane = threadIdx.x & 63u, leader = (uint32_t)__ffsll((long long)gm) - 1u, li = blockIdx.x & (GEN_LISTS - 1u);
            uint32_t base = 0;
            if (lane == leader) base = atomicAdd(gen_cnt + li, (uint32_t)__popcll(gm));
            base = (uint32_t)__shfl((int)base, (int)leader, 64);
            if (to_gen) gen_list[(uint64_t)li * gen_cap + base + (uint32_t)__popcll(gm & ((1ull << lane) - 1ull))] = t;
        }
    }
    // anchors of the pair: one atomic per wavefront when all its chunks belong to one pair (nearly always)
    const uint32_t pi0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)pi);
    if (__all(!in || pi == pi0)) {
        uint32_t v = n_add;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if ((threadIdx.x & 63u) == 0 && v) atomicAdd(&pair_na[pi0], v);
    } else if (n_add) {
        atomicAdd(&pair_na[pi], n_add);
    }
}

// ---------------------------------------------------------------------------------------------
// FAST PATH on runs: one lane per (pair, 20 kb chunk).
//
// The lane reads the run records of its chunk (contiguous, in seed order, from chunk_rec0 on) and runs the
// banded chaining DP of ani_oracle.c on a compressed
// state: a register ring of the 4 most recently touched RUNS.  A run is a stretch of anchors each chained to
// the one before at a gap cost of at most RUN_GAP (same record and strand, diagonal steps <= 10); its scores
// rise by at least 10 per anchor while its diagonal moves by at most 10, so among the anchors of a run only
// the last one can be the best predecessor of a later anchor (it is nearer and offers at least as much) --
// unless the later anchor lies inside the run's own extent, which is detected and declined.
// Every record is ONE step: its first anchor walks the ring exactly like the oracle's look-back loop
// (nearest first, strict '>', early exits on the running maximum / 2500-base band / 50-anchor band); its
// other n - 1 anchors follow at once when no other run or summary can offer any of them more than the run
// itself does -- per anchor the run's own offer rises by 20 - gap >= 10 and an offer from elsewhere (score +
// 20 - diagonal distance) by at most the gap <= 10, so the test at the second anchor covers all of them
// (equal offers go to the nearest anchor, the run's own previous one).  Runs that fall out of the ring are kept as summaries (best
// score, last position, diagonal range); a look-back that would have to continue into them is accepted only
// if no summarised anchor can reach the current best.  The lane proves as it goes that its result is the
// oracle's; a chunk where the proof fails (branching chains, best end not last, too many hits or chains)
// goes to the slow path.  Rounds are uniform across the wavefront: every live lane takes one record per round.
struct Run {
    uint32_t q_last, rr_last;         // last anchor: query pos; hit word (ref pos | record tag << 24 | rev << 31)
    int32_t f;                        // score of the last anchor
    uint32_t cnt;                     // anchors on the PATH ending at the last anchor | SUCC_BIT
    uint32_t first_qi, q_first, r_pfirst;     // path aggregates: first seed index and its position, ref extent
    uint32_t qi_last, idx_last;       // seed index / anchor ordinal of the last anchor
    int32_t pmax;                     // highest score among the earlier anchors of the path
    uint32_t r_first;                 // ref pos of the run's first anchor
    uint32_t seg;                     // summary key: changes along a path only at score-lowering indels
    int32_t gs;                       // diagonal steps inside the run: its earlier anchors lie at most this far off the last one's diagonal
};

__global__ __launch_bounds__(256) void chain_runs_kernel(SetView A, SetView B, const PairDesc *__restrict__ pairs, uint32_t npairs,
                                                         const uint32_t *__restrict__ gen_list, const uint32_t *__restrict__ gen_cnt,
                                                         uint32_t gen_cap, const RunRec *__restrict__ recs,
                                                         const uint32_t *__restrict__ chunk_rec0, const uint4 *__restrict__ multi,
                                                         ChainRec *__restrict__ fast_chains, uint32_t *__restrict__ chunk_state,
                                                         uint32_t *__restrict__ slow_list, uint32_t *__restrict__ slow_count,
                                                         uint32_t *__restrict__ pair_na, const uint32_t *__restrict__ chunk_pair)
{
    // the chunks chain_single_kernel could not settle, one per lane; their number is only known on the device: a fixed
    // grid strides over the GEN_LISTS lists laid end to end (offsets by a scan of the 256 counts, in LDS)
    __shared__ uint32_t g_off[GEN_LISTS + 1], g_ws[4];
    {
        uint32_t total;
        const uint32_t ex = block_excl_scan_256(gen_cnt[threadIdx.x], g_ws, total);
        g_off[threadIdx.x] = ex;
        if (threadIdx.x == 0) g_off[GEN_LISTS] = total;
        __syncthreads();
    }
    const uint32_t n_items = g_off[GEN_LISTS];
    if (blockIdx.x == 0 && threadIdx.x == 0) slow_count[11] = n_items;      // for the host's statistics
    for (uint32_t w0 = blockIdx.x * 256u; w0 < n_items; w0 += gridDim.x * 256u) {
    const uint32_t w = w0 + threadIdx.x;
    const bool live = w < n_items;
    uint32_t t = 0;
    if (live) {
        uint32_t lo = 0, hi = GEN_LISTS;              // the list that holds item w: last offset <= w
        while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (g_off[mid] <= w) lo = mid; else hi = mid; }
        t = gen_list[(uint64_t)lo * gen_cap + (w - g_off[lo])];
    }
    const uint32_t pi = live ? chunk_pair[t] : 0u;        // (chain_single_kernel left it there: a binary search over the pairs is 15 dependent loads)
    const PairDesc pd = pairs[pi];
    const uint32_t idx0 = chunk_rec0[t];
    const uint32_t c = t - pd.chunk_base;
    const SetView &QS = (pd.flags & 2u) ? B : A;
    const uint32_t s0 = QS.chunk_start[pd.q_chunk_off + c], s1 = QS.chunk_start[pd.q_chunk_off + c + 1];
    bool cplx = false;
    uint32_t cause = 0u;

    const int32_t NEG = -0x40000000;
    Run r0, r1, r2, r3;
    r0.cnt = r1.cnt = r2.cnt = r3.cnt = 0;          // cnt == 0: empty ring position
    r0.f = r1.f = r2.f = r3.f = NEG;
    r0.q_last = r1.q_last = r2.q_last = r3.q_last = 0; r0.rr_last = r1.rr_last = r2.rr_last = r3.rr_last = 0;
    r0.first_qi = r1.first_qi = r2.first_qi = r3.first_qi = 0; r0.q_first = r1.q_first = r2.q_first = r3.q_first = 0;
    r0.r_pfirst = r1.r_pfirst = r2.r_pfirst = r3.r_pfirst = 0;
    r0.qi_last = r1.qi_last = r2.qi_last = r3.qi_last = 0; r0.idx_last = r1.idx_last = r2.idx_last = r3.idx_last = 0;
    r0.pmax = r1.pmax = r2.pmax = r3.pmax = NEG; r0.r_first = r1.r_first = r2.r_first = r3.r_first = 0;
    r0.seg = r1.seg = r2.seg = r3.seg = 0;
    r0.gs = r1.gs = r2.gs = r3.gs = 0;
    uint32_t ia = 0, nfin = 0, nevict = 0;
    int32_t runmax = NEG;
    // summaries of runs that left the ring: the most recent segment, plus one conservative scalar
    uint32_t s0_seg = 0xFFFFFFFFu, s0_key = 0, s0_q = 0, lost_q = 0;
    int32_t s0_f = NEG, lost_f = NEG, s0_dlo = 0, s0_dhi = 0, lost_dlo = 0, lost_dhi = 0;
    // the keyless summary keeps TWO diagonal intervals (empty: lo > hi): the remnants of the main path and a
    // stray single hit far off its diagonal would otherwise merge into one interval that covers everything in between
    int32_t lost2_dlo = 1, lost2_dhi = 0;
    ChainRec *slots = fast_chains + (uint64_t)t * FAST_SLOTS;

#define EMIT_PATH(E)                                                                         \
    do {                                                                                     \
        if ((E).cnt && !((E).cnt & SUCC_BIT) && (E).cnt >= ANI_MIN_ANCHORS) {                \
            if (!((E).f > (E).pmax)) { cplx = true; cause = 5; } /* best end is not the last anchor */ \
            else if (nfin >= FAST_SLOTS) { cplx = true; cause = 1; }                         \
            else {                                                                           \
                ChainRec cr;                                                                 \
                cr.score = (E).f; cr.n = (E).cnt; cr.n_seeds = (E).qi_last - (E).first_qi + 1; \
                cr.q0 = (E).q_first; cr.q1 = (E).q_last;                                       \
                { /* a predecessor lies strictly behind on the other genome too: the path's extent there is spanned by its two ends */ \
                  const uint32_t rl_ = (E).rr_last & HIT_POS_MASK;                            \
                  cr.r0 = rl_ < (E).r_pfirst ? rl_ : (E).r_pfirst; cr.r1 = rl_ > (E).r_pfirst ? rl_ : (E).r_pfirst; } \
                cr.chunk = c; \
                slots[nfin++] = cr;                                                          \
            }                                                                                \
        }                                                                                    \
    } while (0)

    // a run leaves the ring: it can no longer be extended; fold it into the summaries
#define EVICT(E)                                                                             \
    do {                                                                                     \
        if ((E).cnt) {                                                                       \
            EMIT_PATH(E);                                                                    \
            nevict++;                                                                        \
            const uint32_t k3 = (E).rr_last & HIT_KEY_MASK;                                   \
            const int32_t d3 = ((E).rr_last >> 31) ? (int32_t)((E).rr_last & HIT_POS_MASK) + (int32_t)(E).q_last \
                                                   : (int32_t)((E).rr_last & HIT_POS_MASK) - (int32_t)(E).q_last; \
            if ((E).seg == s0_seg) {                                                         \
                s0_f = (E).f > s0_f ? (E).f : s0_f; s0_q = (E).q_last > s0_q ? (E).q_last : s0_q; \
                s0_dlo = d3 - (E).gs < s0_dlo ? d3 - (E).gs : s0_dlo; s0_dhi = d3 + (E).gs > s0_dhi ? d3 + (E).gs : s0_dhi; \
            } else {                                                                         \
                if (s0_seg != 0xFFFFFFFFu) {                                                 \
                    if (lost_f == NEG) { lost_dlo = s0_dlo; lost_dhi = s0_dhi; }             \
                    else {                                                                   \
                        const int32_t g1a = s0_dlo - lost_dhi, g1b = lost_dlo - s0_dhi;      \
                        const int32_t g1 = g1a > g1b ? (g1a > 0 ? g1a : 0) : (g1b > 0 ? g1b : 0); /* distance to interval 1 */ \
                        bool into1 = g1 <= 2 * ANI_MAX_GAP;                                  \
                        if (!into1 && lost2_dlo <= lost2_dhi) {                              \
                            const int32_t g2a = s0_dlo - lost2_dhi, g2b = lost2_dlo - s0_dhi; \
                            const int32_t g2 = g2a > g2b ? (g2a > 0 ? g2a : 0) : (g2b > 0 ? g2b : 0); \
                            into1 = g1 <= g2;                                                \
                            if (!into1) { lost2_dlo = s0_dlo < lost2_dlo ? s0_dlo : lost2_dlo; lost2_dhi = s0_dhi > lost2_dhi ? s0_dhi : lost2_dhi; } \
                        } else if (!into1) { lost2_dlo = s0_dlo; lost2_dhi = s0_dhi; }        \
                        if (into1) { lost_dlo = s0_dlo < lost_dlo ? s0_dlo : lost_dlo; lost_dhi = s0_dhi > lost_dhi ? s0_dhi : lost_dhi; } \
                    }                                                                        \
                    lost_f = s0_f > lost_f ? s0_f : lost_f; lost_q = s0_q > lost_q ? s0_q : lost_q; \
                }                                                                            \
                s0_seg = (E).seg; s0_key = k3; s0_f = (E).f; s0_q = (E).q_last; s0_dlo = d3 - (E).gs; s0_dhi = d3 + (E).gs; \
            }                                                                                \
        }                                                                                    \
    } while (0)

    // record cursor: the chunk's records follow one another in the pair's region, from chunk_rec0 on, in seed order; the
    // record BEHIND a run closes it (a link or the terminator at the end of a quarter), so two records are held and
    // the third is on its way while the first is worked on
    const uint4 *prec = reinterpret_cast<const uint4 *>(recs + pd.rec_base);
    uint32_t idx = idx0;
    bool done = !live || idx == 0xFFFFFFFFu || s1 <= s0;
    uint4 a0 = make_uint4(0, 0, 0, 0), a1 = a0, b0 = a0, b1 = a0;
    if (!done) { a0 = prec[2u * idx]; a1 = prec[2u * idx + 1u]; b0 = prec[2u * idx + 2u]; b1 = prec[2u * idx + 3u]; }   // a run record is never the last of its quarter
    // One ANCHOR per round and lane: a record's first anchor -- or, for a seed with 2..4 occurrences on the other genome, one of
    // its occurrences per round (pend = occurrences still to come; the lane keeps its record until they are through).  A loop
    // over the occurrences inside the round made the whole wavefront repeat the look-back as often as its most repetitive seed
    // asked: on real genome structure 29 % of the records are such seeds and 82 % of the rounds had one in some lane.
    struct { uint32_t qi, q0, hw, q1, qi1, hw1, n, gsum; } rc;
    rc.qi = 0; rc.q0 = 0; rc.hw = HIT_NONE; rc.hw1 = HIT_NONE; rc.q1 = 0; rc.qi1 = 0; rc.n = 0; rc.gsum = 0;
    uint32_t pend = 0, g0 = HIT_NONE, g1 = HIT_NONE, g2 = HIT_NONE, g3 = HIT_NONE;
    for (;;) {
        bool have = pend != 0u;
        if (!done && !have) {
            if (a0.x == REC_LINK && s1 <= a0.z) done = true;       // the chunk ends with its quarter
            else if (a0.x == REC_LINK) {                  // the chunk goes on in the next quarter of the region
                idx = a0.y;
                a0 = prec[2u * idx]; a1 = prec[2u * idx + 1u];
                if (a0.x < REC_LINK) { b0 = prec[2u * idx + 2u]; b1 = prec[2u * idx + 3u]; }
            } else if (a0.x >= s1) done = true;           // records are in seed order (terminator: ~0): the chunk is finished
            else {
                have = true;
                rc.qi = a0.x; rc.q0 = a0.y; rc.hw = a0.z;
                rc.n = b0.w - a0.w; rc.gsum = b1.w - a1.w;                 // running counts: this run's share
                rc.q1 = b1.x; rc.hw1 = b1.y; rc.qi1 = b1.z;               // the hit in front of the next record ends this run
                a0 = b0; a1 = b1;
                idx++;
                if (a0.x < REC_LINK) { b0 = prec[2u * idx + 2u]; b1 = prec[2u * idx + 3u]; }
            }
        }
        if (have) do {
            // ---- one anchor of the record (its first, or the next occurrence of a multi-occurrence seed) through the look-back
            const uint32_t s = rc.qi;
            const int32_t qp = (int32_t)rc.q0;
            const uint32_t hw = rc.hw;
            if (pend == 0u) {            // a new record
                if (hw == HIT_MANY) { cplx = true; cause = 2; break; }
                pend = 1u; g0 = hw; g1 = g2 = g3 = HIT_NONE;
                if ((hw & 0xFF000000u) == HIT_MULTI) {   // 2..4 occurrences, ascending gpos
                    const uint4 mv = multi[pd.multi_base + (hw & 0x00FFFFFFu)];
                    g0 = mv.x; g1 = mv.y; g2 = mv.z; g3 = mv.w;
                    pend = 2u + (g2 != HIT_NONE) + (g3 != HIT_NONE);
                }
            }
            {
                const uint32_t rr = g0;
                g0 = g1; g1 = g2; g2 = g3;
                pend--;
                const int32_t rp = (int32_t)(rr & HIT_POS_MASK);
                const uint32_t rev = rr >> 31;
                const uint32_t key = rr & HIT_KEY_MASK;     // strand + record tag
                const int32_t dg = rev ? rp + qp : rp - qp;
                // ---- the oracle's look-back over the last anchors of the ring's runs
                int32_t best = ANI_ANCHOR_SCORE, pgap = 0;
                int bj = -1;
                bool exact = false;
    #define TRY(K, E)                                                                                   \
                if (!exact && !cplx) {                                                                  \
                    if (!(E).cnt) exact = true;                       /* no older anchors at all */     \
                    else if (best >= runmax + ANI_ANCHOR_SCORE) exact = true;                           \
                    else if (ia - (E).idx_last > ANI_BAND) exact = true;                                \
                    else {                                                                              \
                        const int32_t dq = qp - (int32_t)(E).q_last;                                    \
                        if (dq > ANI_BP_BAND) exact = true;                                             \
                        else if (((E).rr_last & HIT_KEY_MASK) == key) {                                 \
                            const int32_t rpj = (int32_t)((E).rr_last & HIT_POS_MASK);                   \
                            const int32_t dr = rev ? rpj - rp : rp - rpj;                               \
                            const int32_t ed = rev ? rpj + (int32_t)(E).q_last : rpj - (int32_t)(E).q_last; /* run diagonal */ \
                            const int32_t off = dg > ed ? dg - ed : ed - dg;                            \
                            /* an earlier anchor of a run with steps may be in reach where the last one is not */ \
                            if (off > ANI_MAX_GAP && off - (E).gs <= ANI_MAX_GAP) { cplx = true; cause = 7; } \
                            else if (off <= ANI_MAX_GAP) {                                              \
                                /* an INTERIOR anchor of the run could be a valid predecessor where the last one is not */ \
                                const int32_t rf = (int32_t)(E).r_first;                                \
                                const bool inside = rev ? (rp < rf && dr <= 0) : (rp > rf && dr <= 0);  \
                                if (dq <= 0 || inside) { cplx = true; cause = 7; }                      \
                                else if (dr > 0 && dq <= ANI_MAX_LIN && dr <= ANI_MAX_LIN) {            \
                                    const int32_t sc = (E).f + ANI_ANCHOR_SCORE - off;                  \
                                    if (sc > best) { best = sc; bj = (K); pgap = off; }                 \
                                }                                                                       \
                            }                                                                           \
                        }                                                                               \
                    }                                                                                   \
                }
                TRY(0, r0) TRY(1, r1) TRY(2, r2) TRY(3, r3)
    #undef TRY
                if (cplx) break;
                if (!exact && nevict) {
                    // the look-back would continue into evicted runs: accept only if none of them can matter
                    bool ok = true;
    #define SUMMARY_BLOCKS(SF, SQ, DLO, DHI, KEYOK)                                                                  \
                    if ((KEYOK) && qp - (int32_t)(SQ) <= ANI_BP_BAND) {                                              \
                        const int32_t off = dg < (DLO) ? (DLO) - dg : (dg > (DHI) ? dg - (DHI) : 0);                  \
                        if (off <= ANI_MAX_GAP && !(best >= (SF) + ANI_ANCHOR_SCORE - off)) ok = false;               \
                    }
                    SUMMARY_BLOCKS(s0_f, s0_q, s0_dlo, s0_dhi, s0_seg != 0xFFFFFFFFu && s0_key == key)
                    SUMMARY_BLOCKS(lost_f, lost_q, lost_dlo, lost_dhi, lost_f != NEG)
                    SUMMARY_BLOCKS(lost_f, lost_q, lost2_dlo, lost2_dhi, lost_f != NEG && lost2_dlo <= lost2_dhi)
    #undef SUMMARY_BLOCKS
                    if (!ok) { cplx = true; cause = 3; break; }
                }
                if (bj >= 0 && pgap == 0) {
                    // same diagonal: the predecessor run comes to the front of the ring and grows
                    if (bj == 1) { const Run tr = r1; r1 = r0; r0 = tr; }
                    else if (bj == 2) { const Run tr = r2; r2 = r1; r1 = r0; r0 = tr; }
                    else if (bj == 3) { const Run tr = r3; r3 = r2; r2 = r1; r1 = r0; r0 = tr; }
                    if (r0.cnt & SUCC_BIT) { cplx = true; cause = 4; break; }   // two anchors chain to one predecessor
                    r0.pmax = r0.f > r0.pmax ? r0.f : r0.pmax;
                    r0.f = best;
                    r0.q_last = (uint32_t)qp; r0.rr_last = rr; r0.cnt += 1u;
                    r0.qi_last = s; r0.idx_last = ia;
                } else {
                    Run e;
                    e.q_last = (uint32_t)qp; e.rr_last = rr; e.f = best;
                    e.qi_last = s; e.idx_last = ia; e.r_first = (uint32_t)rp; e.gs = 0;
                    if (bj >= 0) {
                        // an indel: new run on the same path; the old run's last anchor now has a successor.  The predecessor
                        // run STAYS where it is: the ring is ordered by the LAST ANCHOR of its runs (the look-back stops at the
                        // first run beyond a band and trusts that older ones, in the ring and in the summaries, lie further
                        // back), and this run's last anchor did not move -- only its fields are read and its mark is set
                        const uint32_t pc = bj == 0 ? r0.cnt : bj == 1 ? r1.cnt : bj == 2 ? r2.cnt : r3.cnt;
                        if (pc & SUCC_BIT) { cplx = true; cause = 4; break; }   // two anchors chain to one predecessor
                        const int32_t pf = bj == 0 ? r0.f : bj == 1 ? r1.f : bj == 2 ? r2.f : r3.f;
                        const int32_t pp = bj == 0 ? r0.pmax : bj == 1 ? r1.pmax : bj == 2 ? r2.pmax : r3.pmax;
                        e.cnt = pc + 1u;
                        e.first_qi = bj == 0 ? r0.first_qi : bj == 1 ? r1.first_qi : bj == 2 ? r2.first_qi : r3.first_qi;
                        e.q_first = bj == 0 ? r0.q_first : bj == 1 ? r1.q_first : bj == 2 ? r2.q_first : r3.q_first;
                        e.r_pfirst = bj == 0 ? r0.r_pfirst : bj == 1 ? r1.r_pfirst : bj == 2 ? r2.r_pfirst : r3.r_pfirst;
                        e.pmax = pf > pp ? pf : pp;
                        e.seg = pgap >= ANI_ANCHOR_SCORE ? ia : (bj == 0 ? r0.seg : bj == 1 ? r1.seg : bj == 2 ? r2.seg : r3.seg);
                        if (bj == 0) r0.cnt |= SUCC_BIT; else if (bj == 1) r1.cnt |= SUCC_BIT; else if (bj == 2) r2.cnt |= SUCC_BIT; else r3.cnt |= SUCC_BIT;
                    } else {
                        e.cnt = 1; e.first_qi = s; e.q_first = (uint32_t)qp; e.r_pfirst = (uint32_t)rp;
                        e.pmax = NEG; e.seg = ia;
                    }
                    EVICT(r3);
                    r3 = r2; r2 = r1; r1 = r0; r0 = e;
                }
                ia++;
                runmax = best > runmax ? best : runmax;
            }
            if (cplx) break;
            if (pend == 0u && rc.n > 1u) {
                // ---- the run's other anchors: extensions of r0 (which holds the anchor just placed) along the run,
                // provided nothing else can offer its second anchor more than r0 does (header comment): every other
                // run / summary is empty, of another record or strand, beyond the 2500-base band already at the first
                // anchor, further off than max_gap plus all the diagonal steps of the run, or scores no more than
                // r0.f + its diagonal distance (- 20 when the run has steps)
                const uint32_t k0 = hw & HIT_KEY_MASK;
                const int32_t rp0 = (int32_t)(hw & HIT_POS_MASK);
                const int32_t d0 = (hw >> 31) ? rp0 + qp : rp0 - qp;
                const int32_t G = (int32_t)rc.gsum, slack = G ? 2 * RUN_GAP : 0;
                const int32_t f0 = r0.f - slack;
    #define DIAG_OFF(E) abs((((E).rr_last >> 31) ? (int32_t)((E).rr_last & HIT_POS_MASK) + (int32_t)(E).q_last               \
                                                 : (int32_t)((E).rr_last & HIT_POS_MASK) - (int32_t)(E).q_last) - d0)
    #define HARMLESS(E)                                                                                   \
                (!(E).cnt || ((E).rr_last & HIT_KEY_MASK) != k0 || qp - (int32_t)(E).q_last > ANI_BP_BAND ||    \
                 DIAG_OFF(E) - G - (E).gs > ANI_MAX_GAP || (E).f - DIAG_OFF(E) <= f0)
                bool domr = !(r0.cnt & SUCC_BIT) && HARMLESS(r1) && HARMLESS(r2) && HARMLESS(r3);
    #undef HARMLESS
    #undef DIAG_OFF
                if (domr && s0_seg != 0xFFFFFFFFu && s0_key == k0 && qp - (int32_t)s0_q <= ANI_BP_BAND) {
                    const int32_t off = d0 < s0_dlo ? s0_dlo - d0 : (d0 > s0_dhi ? d0 - s0_dhi : 0);
                    if (off - G <= ANI_MAX_GAP && s0_f - off > f0) domr = false;
                }
                if (domr && lost_f != NEG && qp - (int32_t)lost_q <= ANI_BP_BAND) {
                    const int32_t off1 = d0 < lost_dlo ? lost_dlo - d0 : (d0 > lost_dhi ? d0 - lost_dhi : 0);
                    if (off1 - G <= ANI_MAX_GAP && lost_f - off1 > f0) domr = false;
                    if (lost2_dlo <= lost2_dhi) {
                        const int32_t off2 = d0 < lost2_dlo ? lost2_dlo - d0 : (d0 > lost2_dhi ? d0 - lost2_dhi : 0);
                        if (off2 - G <= ANI_MAX_GAP && lost_f - off2 > f0) domr = false;
                    }
                }
                if (!domr) { cplx = true; cause = 9; break; }
                const uint32_t ext = rc.n - 1u;
                r0.q_last = rc.q1;
                r0.rr_last = rc.hw1;
                r0.f = r0.f + ANI_ANCHOR_SCORE * (int32_t)ext - G;
                // the second-to-last anchor of the run scores at most r0.f - (20 - RUN_GAP): an upper bound serves pmax
                r0.pmax = r0.f - (ANI_ANCHOR_SCORE - RUN_GAP) > r0.pmax ? r0.f - (ANI_ANCHOR_SCORE - RUN_GAP) : r0.pmax;
                runmax = r0.f > runmax ? r0.f : runmax;
                r0.cnt += ext;
                r0.idx_last = ia + ext - 1u; ia += ext;
                r0.qi_last = rc.qi1;
                r0.gs += G;
            }
        } while (0);
        if (cplx) { done = true; pend = 0u; }
#ifdef SKDER_RUNS_STATS
        {   // lanes with a record this round / lanes still at work, per wavefront round; rounds with a multi-occurrence seed
            const unsigned long long hm = __ballot(have), lm = __ballot(!done);
            const bool is_multi = have && (rc.hw & 0xFF000000u) == HIT_MULTI && rc.hw != HIT_MANY;
            const unsigned long long mm = __ballot(is_multi);
            if ((threadIdx.x & 63u) == 0) { atomicAdd(slow_count + 12, 1u); atomicAdd(slow_count + 13, (uint32_t)__popcll(hm)); atomicAdd(slow_count + 14, (uint32_t)__popcll(lm));
                                            if (mm) atomicAdd(slow_count + 17, 1u); atomicAdd(slow_count + 18, (uint32_t)__popcll(mm)); }
        }
#endif
        if (!__any(!done || pend != 0u)) break;     // the whole wave is finished
    }
    if (!cplx) EMIT_PATH(r3);
    if (!cplx) EMIT_PATH(r2);
    if (!cplx) EMIT_PATH(r1);
    if (!cplx) EMIT_PATH(r0);
#undef EMIT_PATH
#undef EVICT
    if (live) {
        if (cplx) {
            chunk_state[t] = CHUNK_SLOW;
            slow_list[atomicAdd(slow_count, 1u)] = t;
            atomicAdd(slow_count + 1 + cause, 1u);
        } else {
            chunk_state[t] = nfin;
            if (ia) atomicAdd(&pair_na[pi], ia);
        }
    }
    }   // items of this lane
}

// ---------------------------------------------------------------------------------------------
// SLOW PATH (unabridged algorithm) for the chunks the fast path declined

// one wavefront per slow chunk: the exact number of anchors of the chunk (a seed may occur any number of
// times on the other genome as long as the repetitive cut-off is inactive, so no a-priori bound exists)
__global__ __launch_bounds__(256) void slow_caps_kernel(SetView A, SetView B, const PairDesc *__restrict__ pairs, uint32_t npairs,
                                                        const uint32_t *__restrict__ slow_list, uint32_t nslow,
                                                        uint32_t *__restrict__ cap)
{
    const uint32_t w = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (w > nslow) return;
    if (w == nslow) { if (lane == 0) cap[w] = 0; return; }
    const uint32_t t = slow_list[w];
    const PairDesc pd = pairs[find_pair(pairs, npairs, t)];
    const SetView &QS = (pd.flags & 2u) ? B : A;
    const SetView &RS = (pd.flags & 4u) ? B : A;
    const GenomeMeta Q = QS.meta[pd.q], R = RS.meta[pd.r];
    const uint32_t c = t - pd.chunk_base;
    const uint32_t s0 = QS.chunk_start[Q.chunk_off + c], s1 = QS.chunk_start[Q.chunk_off + c + 1];
    const uint32_t *qk = QS.pkmer + Q.seed_off;
    const uint32_t *rk = RS.skmer + R.seed_off, *rb = RS.boff + R.bucket_off;
    const uint32_t *qsk = QS.skmer + Q.seed_off, *qb = QS.boff + Q.bucket_off;
    uint32_t mine = 0;
    for (uint32_t s = s0 + lane; s < s1; s += 64) {
        const uint32_t kmer = qk[s] & SK_SEED_MASK;
        const uint32_t b = kmer_bucket(kmer, R.bucket_bits);
        uint32_t cnt = 0;
        for (uint32_t e = rb[b]; e < rb[b + 1]; e++) {
            const uint32_t k2 = rk[e] & SK_SEED_MASK;
            if (k2 == kmer) cnt++;
            else if (k2 > kmer) break;
        }
        if (cnt > R.rep_cut) cnt = 0;
        if (cnt && Q.rep_cut != 0xFFFFFFFFu) {
            const uint32_t b2 = kmer_bucket(kmer, Q.bucket_bits);
            uint32_t m2 = 0;
            for (uint32_t e = qb[b2]; e < qb[b2 + 1]; e++) m2 += ((qsk[e] & SK_SEED_MASK) == kmer);
            if (m2 > Q.rep_cut) cnt = 0;
        }
        mine += cnt;
    }
    uint32_t total;
    (void)wave_excl_scan(mine, total);
    if (lane == 0) cap[w] = total;
}

// one wavefront per slow chunk: ordered anchors through the bucket index (hits in ascending gpos)
__global__ __launch_bounds__(256) void slow_anchors_kernel(SetView A, SetView B, const PairDesc *__restrict__ pairs, uint32_t npairs,
                                                           const uint32_t *__restrict__ slow_list, uint32_t nslow,
                                                           const uint32_t *__restrict__ abase, uint32_t *__restrict__ a_qi,
                                                           uint32_t *__restrict__ a_r, uint32_t *__restrict__ a_rctg,
                                                           uint32_t *__restrict__ slow_n, uint32_t *__restrict__ flags)
{
    const uint32_t w = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (w >= nslow) return;
    const uint32_t t = slow_list[w];
    const PairDesc pd = pairs[find_pair(pairs, npairs, t)];
    const SetView &QS = (pd.flags & 2u) ? B : A;
    const SetView &RS = (pd.flags & 4u) ? B : A;
    const GenomeMeta Q = QS.meta[pd.q], R = RS.meta[pd.r];
    const uint32_t c = t - pd.chunk_base;
    const uint32_t s0 = QS.chunk_start[Q.chunk_off + c], s1 = QS.chunk_start[Q.chunk_off + c + 1];
    const uint32_t *qk = QS.pkmer + Q.seed_off;
    const uint32_t *rk = RS.skmer + R.seed_off, *rg = RS.sgpos + R.seed_off, *rc = RS.sctg + R.seed_off;
    const uint32_t *rb = RS.boff + R.bucket_off;
    const uint32_t *qsk = QS.skmer + Q.seed_off, *qb = QS.boff + Q.bucket_off;
    const uint32_t base_out = abase[w], cap = abase[w + 1] - abase[w];
    uint32_t running = 0;
    for (uint32_t sb = s0; sb < s1; sb += 64) {
        const uint32_t s = sb + lane;
        uint32_t cnt = 0, first = 0, km = 0;
        if (s < s1) {
            km = qk[s];
            const uint32_t kmer = km & SK_SEED_MASK;
            const uint32_t b = kmer_bucket(kmer, R.bucket_bits);
            const uint32_t lo = rb[b], hi = rb[b + 1];
            for (uint32_t e = lo; e < hi; e++) {
                const uint32_t k2 = rk[e] & SK_SEED_MASK;
                if (k2 == kmer) { if (!cnt) first = e; cnt++; }
                else if (k2 > kmer) break;
            }
            if (cnt > R.rep_cut) cnt = 0;
            if (cnt && Q.rep_cut != 0xFFFFFFFFu) {   // multiplicity inside the chunked genome itself
                const uint32_t b2 = kmer_bucket(kmer, Q.bucket_bits);
                uint32_t m2 = 0;
                for (uint32_t e = qb[b2]; e < qb[b2 + 1]; e++) m2 += ((qsk[e] & SK_SEED_MASK) == kmer);
                if (m2 > Q.rep_cut) cnt = 0;
            }
        }
        uint32_t total;
        const uint32_t at = running + wave_excl_scan(cnt, total);
        if (cnt) {
            if (at + cnt <= cap) {
                for (uint32_t u = 0; u < cnt; u++) {
                    const uint32_t idx = base_out + at + u;
                    const uint32_t rkm = rk[first + u];
                    a_qi[idx] = s;
                    a_r[idx] = rg[first + u] | (((km >> 31) != (rkm >> 31)) ? USED_BIT : 0u);
                    a_rctg[idx] = rc[first + u];
                }
            } else {
                atomicOr(&flags[0], 4u);
            }
        }
        running += total;
    }
    if (lane == 0) slow_n[w] = running < cap ? running : cap;
}

// forward declaration (defined with the wave kernel below)
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v);

// one WAVEFRONT per chunk of the over list; anchors in global memory.  The DP only ever looks back 50
// anchors, so the last 64 anchors are kept in a per-wave LDS ring and the 64 lanes examine the look-back
// candidates of one anchor together (as slow_wave_kernel does); f and bp go to global memory for the
// extraction, which lane 0 runs over the candidate ends sorted once.
__global__ __launch_bounds__(256) void slow_chain_kernel(SetView A, SetView B, const PairDesc *__restrict__ pairs, uint32_t npairs,
                                                         const uint32_t *__restrict__ slow_list, uint32_t nslow,
                                                         const uint32_t *__restrict__ abase, const uint32_t *__restrict__ slow_n,
                                                         const uint32_t *__restrict__ a_qi, const uint32_t *__restrict__ a_r,
                                                         const uint32_t *__restrict__ a_rctg, int32_t *__restrict__ F,
                                                         uint32_t *__restrict__ BP, uint64_t *__restrict__ ORD, ChainRec *__restrict__ chains,
                                                         uint32_t *__restrict__ pair_nch, uint32_t *__restrict__ pair_na,
                                                         uint32_t *__restrict__ flags)
{
    __shared__ uint32_t ring_qp[4][64], ring_rr[4][64], ring_rc[4][64];
    __shared__ int32_t ring_f[4][64];
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t w = blockIdx.x * 4u + wv;
    if (w >= nslow) return;
    const uint32_t t = slow_list[w];
    const uint32_t lo = find_pair(pairs, npairs, t);
    const PairDesc pd = pairs[lo];
    const uint32_t a0 = abase[w], n = slow_n[w];
    if (!n) return;
    if (lane == 0) atomicAdd(&pair_na[lo], n);
    const SetView &QS = (pd.flags & 2u) ? B : A;
    const uint32_t *qg = QS.pgpos + QS.meta[pd.q].seed_off;
    const uint32_t *qi = a_qi + a0, *ar = a_r + a0, *ac = a_rctg + a0;
    int32_t *f = F + a0;
    uint32_t *bp = BP + a0;

    // banded chaining: lane l examines predecessor i-1-l of anchor i (ring slot (i-1-l) & 63)
    for (uint32_t i = 0; i < n; i++) {
        const int32_t qpi = (int32_t)qg[qi[i]];
        const uint32_t rr = ar[i], rc = ac[i];
        const int32_t rp = (int32_t)(rr & 0x7FFFFFFFu);
        const uint32_t rev = rr >> 31;
        uint32_t key = 0;     // (score << 6) | (63 - lane): the maximum is the best score, nearest on ties
        if (lane < i && lane < ANI_BAND) {
            const uint32_t sl = (i - 1 - lane) & 63u;
            const int32_t dq = qpi - (int32_t)ring_qp[wv][sl];
            const uint32_t rj = ring_rr[wv][sl];
            if (dq <= ANI_BP_BAND && ring_rc[wv][sl] == rc && (rj >> 31) == rev) {
                const int32_t rpj = (int32_t)(rj & 0x7FFFFFFFu);
                const int32_t dr = rev ? rpj - rp : rp - rpj;
                if (dq > 0 && dr > 0 && dq <= ANI_MAX_LIN && dr <= ANI_MAX_LIN) {
                    const int32_t gap = dq > dr ? dq - dr : dr - dq;
                    if (gap <= ANI_MAX_GAP) {
                        const int32_t sc = ring_f[wv][sl] + ANI_ANCHOR_SCORE - gap;
                        if (sc > ANI_ANCHOR_SCORE) key = ((uint32_t)sc << 6) | (63u - lane);
                    }
                }
            }
        }
        key = wave_max_u32(key);
        if (lane == 0) {
            const int32_t fi = key ? (int32_t)(key >> 6) : ANI_ANCHOR_SCORE;
            f[i] = fi;
            bp[i] = key ? i - (63u - (key & 63u)) : 0u;     // predecessor index + 1
            ring_qp[wv][i & 63u] = (uint32_t)qpi; ring_rr[wv][i & 63u] = rr; ring_rc[wv][i & 63u] = rc; ring_f[wv][i & 63u] = fi;
        }
        __builtin_amdgcn_wave_barrier();
    }
    __threadfence();
    if (lane != 0) return;
    // chains: best end first (ties: lowest index); back-track until the start or a used anchor.  Scores
    // never change except by being voided, so the order of the candidate ends is fixed: sort them once
    // (heap sort of (score, ~index) keys, descending) instead of scanning for the maximum per chain --
    // a chunk inside a shared tandem repeat has 10^5 anchors and thousands of chains
    uint64_t *key = ORD + a0;
    uint32_t m = 0;
    for (uint32_t i = 0; i < n; i++)
        if (f[i] > ANI_ANCHOR_SCORE) key[m++] = ((uint64_t)(uint32_t)f[i] << 32) | (uint64_t)(0xFFFFFFFFu - i);
    {
        auto sift = [&](uint32_t root, uint32_t end) {      // min-heap: the array ends up in descending order
            const uint64_t v = key[root];
            for (;;) {
                uint32_t c = 2 * root + 1;
                if (c >= end) break;
                if (c + 1 < end && key[c + 1] < key[c]) c++;
                if (!(key[c] < v)) break;
                key[root] = key[c];
                root = c;
            }
            key[root] = v;
        };
        for (uint32_t i = m / 2; i-- > 0;) sift(i, m);
        for (uint32_t e = m; e-- > 1;) {
            const uint64_t t2 = key[0]; key[0] = key[e]; key[e] = t2;
            sift(0, e);
        }
    }
    for (uint32_t kk = 0; kk < m; kk++) {
        const int32_t besti = (int32_t)(0xFFFFFFFFu - (uint32_t)key[kk]);
        const int32_t bestv = f[besti];
        if (bestv <= ANI_ANCHOR_SCORE) continue;      // voided since: inside an extracted chain, or a failed end
        uint32_t cnt = 0, rmin = 0xFFFFFFFFu, rmax = 0;
        int32_t cur = besti, first = besti;
        while (cur >= 0) {
            const uint32_t b = bp[cur];
            if (b & USED_BIT) break;
            cnt++;
            first = cur;
            const uint32_t rp = ar[cur] & 0x7FFFFFFFu;
            rmin = rp < rmin ? rp : rmin;
            rmax = rp > rmax ? rp : rmax;
            cur = (int32_t)(b & 0x7FFFFFFFu) - 1;
        }
        if (cnt < ANI_MIN_ANCHORS) { f[besti] = (int32_t)0x80000000; continue; }
        cur = besti;
        while (cur >= 0) {
            const uint32_t b = bp[cur];
            if (b & USED_BIT) break;
            bp[cur] = b | USED_BIT;
            f[cur] = (int32_t)0x80000000;
            cur = (int32_t)(b & 0x7FFFFFFFu) - 1;
        }
        const uint32_t slot = atomicAdd(&pair_nch[lo], 1u);
        if (slot < pd.c_cap) {
            ChainRec c;
            c.score = bestv;
            c.n = cnt;
            c.n_seeds = qi[besti] - qi[first] + 1;
            c.q0 = qg[qi[first]];
            c.q1 = qg[qi[besti]];
            c.r0 = rmin; c.r1 = rmax;
            c.chunk = t - pd.chunk_base;
            chains[pd.c_base + slot] = c;
        } else {
            atomicOr(&flags[0], 8u);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// SLOW PATH, main form: one WAVEFRONT per declined chunk, everything in LDS.
// The wave builds the chunk's ordered anchor list through the bucket index (64 seeds at a time, hits
// in ascending gpos), runs the unabridged DP with the 64 lanes spread over the 50 look-back
// candidates of one anchor (packed max-reduce: score first, nearest predecessor on ties), then extracts
// chains best end first with back-tracking by lane 0.  Chunks with more than SLOWW_MAXA anchors are
// passed on to the global-memory kernels above.
//
// The kernel is bound by instruction issue (about 60 wavefront instructions per anchor of the DP, 20 per anchor of the
// back-tracking, one lane busy): on real genomes 10-13 % of the chunks come here and took half of the chain stage.
// LADDERS = true spends those instructions per STRETCH instead of per anchor, with the same result:
//   * DP.  Before the loop every anchor is tested, all in parallel, for "continues the anchor in front of it on the same
//     diagonal": a valid link of gap 0.  At an anchor with that mark whose predecessor holds the highest score so far
//     (f[i-1] == runmax) the look-back is settled without being run: any candidate offers f[j] + 20 - gap <= runmax + 20 =
//     f[i-1] + 20, what the predecessor offers, and ties go to the nearest candidate -- the predecessor.  Its score is then
//     the new maximum, so the argument repeats: the whole stretch of consecutive marks gets f = f[i-1] + 20, 40, ... and
//     bp = the anchor before, in one step (the main path of a chunk: typically 50-100 anchors between two stray hits).
//   * Chains.  Anchors with bp = "the anchor before" form ladders; a chain that enters a ladder takes it down to its
//     bottom, or to the anchors an earlier (better) chain took -- which always form the ladder's lower end, since
//     every chain walks down until it meets used anchors.  One word per ladder (how far up it is used) replaces the used
//     bit per anchor; the walk, the count and the extent on the other genome (monotone along a ladder: its two ends)
//     go ladder by ladder.
#ifndef SLOWW_MAXA
#define SLOWW_MAXA 384
#endif
#define SLOWW_WAVES 4        // wavefronts (chunks) per workgroup

// maximum over the 64 lanes of a fully active wavefront, in every lane: DPP row shifts inside the four
// 16-lane rows, two row broadcasts, one readlane (7 instructions; the shuffle form costs six LDS
// crossbar round trips)
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
    uint32_t t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true); v = t > v ? t : v;   // row_shr:1
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true); v = t > v ? t : v;   // row_shr:2
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true); v = t > v ? t : v;   // row_shr:4
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true); v = t > v ? t : v;   // row_shr:8
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, true); v = t > v ? t : v;   // row_bcast:15 -> rows 1, 3
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, true); v = t > v ? t : v;   // row_bcast:31 -> rows 2, 3
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

template <bool LADDERS>
__global__ __launch_bounds__(64 * SLOWW_WAVES) void slow_wave_kernel(SetView A, SetView B, const PairDesc *__restrict__ pairs, uint32_t npairs,
                                                                    const uint32_t *__restrict__ slow_list, const uint32_t *__restrict__ nslow_ptr,
                                                                    const uint32_t *__restrict__ hits, const uint4 *__restrict__ multi,
                                                                    ChainRec *__restrict__ chains, uint32_t *__restrict__ pair_nch,
                                                                    uint32_t *__restrict__ pair_na, uint32_t *__restrict__ over_list,
                                                                    uint32_t *__restrict__ over_count, uint32_t *__restrict__ flags,
                                                                    const uint32_t *__restrict__ chunk_pair)
{
    __shared__ uint32_t s_qi[SLOWW_WAVES][SLOWW_MAXA], s_qp[SLOWW_WAVES][SLOWW_MAXA], s_rr[SLOWW_WAVES][SLOWW_MAXA];
    __shared__ uint32_t s_rc[SLOWW_WAVES][SLOWW_MAXA], s_bp[SLOWW_WAVES][SLOWW_MAXA];
    __shared__ int32_t s_f[SLOWW_WAVES][SLOWW_MAXA];
    __shared__ unsigned long long s_mask[SLOWW_WAVES][SLOWW_MAXA / 64 + 1];
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    unsigned long long *lmask = s_mask[wv];
    // the number of declined chunks is only known on the device (no host round trip between the fast
    // path and this kernel): a fixed grid strides over the list
    const uint32_t nslow = *nslow_ptr;
    uint32_t *qi = s_qi[wv], *qp = s_qp[wv], *ar = s_rr[wv], *ac = s_rc[wv], *bp = s_bp[wv];
    int32_t *f = s_f[wv];
    for (uint32_t w = blockIdx.x * SLOWW_WAVES + wv; w < nslow; w += gridDim.x * SLOWW_WAVES) {
    __builtin_amdgcn_wave_barrier();
    const uint32_t t = slow_list[w];
    const uint32_t pi = chunk_pair[t];
    const PairDesc pd = pairs[pi];
    const SetView &QS = (pd.flags & 2u) ? B : A;
    const SetView &RS = (pd.flags & 4u) ? B : A;
    const GenomeMeta Q = QS.meta[pd.q], R = RS.meta[pd.r];
    const uint32_t c = t - pd.chunk_base;
    const uint32_t s0 = QS.chunk_start[Q.chunk_off + c], s1 = QS.chunk_start[Q.chunk_off + c + 1];
    const uint32_t *qk = QS.pkmer + Q.seed_off, *qg = QS.pgpos + Q.seed_off;
    const uint32_t *rk = RS.skmer + R.seed_off, *rg = RS.sgpos + R.seed_off, *rcg = RS.sctg + R.seed_off;
    const uint32_t *rb = RS.boff + R.bucket_off;
    const uint32_t *qsk = QS.skmer + Q.seed_off, *qb = QS.boff + Q.bucket_off;

    // 1. ordered anchors.  The join has already found the occurrences of every seed: single hits and the
    // 2..4-occurrence lists are taken from its hit words (one coalesced read per 64 seeds); only seeds
    // marked "too many" -- or all seeds, when the chunked genome's own multiplicity filter is active --
    // are looked up again through the bucket index.  Records are compared by their 6-bit tags, exact
    // under the DP's distance limits like in the fast path.
    // (positions of a genome beyond 2^24 bases do not fit a hit word either: every hit is looked up again)
    const bool qfilter = Q.rep_cut != 0xFFFFFFFFu || R.total_len > (uint64_t)HIT_POS_MASK;
    const bool qrep = Q.rep_cut != 0xFFFFFFFFu;
    const uint32_t *hw_of = hits + pd.hit_base;
    uint32_t n = 0;
    bool over = false;
    for (uint32_t sb = s0; sb < s1; sb += 64) {
        const uint32_t s = sb + lane;
        uint32_t cnt = 0, first = 0, km = 0, hw = HIT_NONE;
        uint4 mv = make_uint4(HIT_NONE, HIT_NONE, HIT_NONE, HIT_NONE);
        bool probe = false;
        if (s < s1) {
            hw = hw_of[s];
            if (qfilter || hw == HIT_MANY) probe = hw != HIT_NONE;
            else if ((hw & 0xFF000000u) == HIT_MULTI) {
                mv = multi[pd.multi_base + (hw & 0x00FFFFFFu)];
                cnt = 2u + (mv.z != HIT_NONE) + (mv.w != HIT_NONE);
            } else if (hw != HIT_NONE) { mv.x = hw; cnt = 1; }
        }
        if (probe) {
            km = qk[s];
            const uint32_t kmer = km & SK_SEED_MASK;
            const uint32_t b = kmer_bucket(kmer, R.bucket_bits);
            const uint32_t lo = rb[b], hi = rb[b + 1];
            for (uint32_t e = lo; e < hi; e++) {
                const uint32_t k2 = rk[e] & SK_SEED_MASK;
                if (k2 == kmer) { if (!cnt) first = e; cnt++; }
                else if (k2 > kmer) break;
            }
            if (cnt > R.rep_cut) cnt = 0;
            if (cnt && qrep) {   // multiplicity inside the chunked genome itself
                const uint32_t b2 = kmer_bucket(kmer, Q.bucket_bits);
                uint32_t m2 = 0;
                for (uint32_t e = qb[b2]; e < qb[b2 + 1]; e++) m2 += ((qsk[e] & SK_SEED_MASK) == kmer);
                if (m2 > Q.rep_cut) cnt = 0;
            }
        }
        uint32_t total;
        const uint32_t at = n + wave_excl_scan(cnt, total);
        if (n + total > SLOWW_MAXA) { over = true; break; }     // wave-uniform
        if (cnt) {
            const uint32_t qpos = qg[s];
            if (probe) {
                for (uint32_t u = 0; u < cnt; u++) {
                    const uint32_t idx = at + u, rkm = rk[first + u];
                    qi[idx] = s; qp[idx] = qpos;
                    ar[idx] = rg[first + u] | (((km >> 31) != (rkm >> 31)) ? USED_BIT : 0u);
                    ac[idx] = rcg[first + u] & 63u;
                }
            } else {
                for (uint32_t u = 0; u < cnt; u++) {
                    const uint32_t idx = at + u, w = u == 0 ? mv.x : (u == 1 ? mv.y : (u == 2 ? mv.z : mv.w));
                    qi[idx] = s; qp[idx] = qpos;
                    ar[idx] = w & (HIT_POS_MASK | USED_BIT);
                    ac[idx] = (w >> HIT_POS_BITS) & 63u;
                }
            }
        }
        n += total;
    }
    if (over) {   // too many anchors for LDS: hand the chunk to the global-memory kernels
        if (lane == 0) over_list[atomicAdd(over_count, 1u)] = t;
        continue;
    }
    if (!n) continue;
    if (lane == 0) atomicAdd(&pair_na[pi], n);
    __builtin_amdgcn_wave_barrier();
#ifdef SKDER_SLOW_STATS
    uint32_t st_full = 0, st_stretch = 0, st_chains = 0, st_walk = 0;
#define SLOW_STAT(X) (X)++
#else
#define SLOW_STAT(X)
#endif

    // 2. banded chaining: lane l examines predecessor i-1-l of anchor i
    if (LADDERS) {
        // marks: anchor i continues anchor i - 1 by a valid link of gap 0 (the conditions of the look-back below, for j = i - 1)
        for (uint32_t b0 = 0; b0 < n; b0 += 64) {
            const uint32_t i = b0 + lane;
            bool ok = false;
            if (i >= 1 && i < n) {
                const uint32_t rr = ar[i], rj = ar[i - 1];
                const int32_t dq = (int32_t)qp[i] - (int32_t)qp[i - 1];
                const int32_t rp = (int32_t)(rr & 0x7FFFFFFFu), rpj = (int32_t)(rj & 0x7FFFFFFFu);
                const int32_t dr = (rr >> 31) ? rpj - rp : rp - rpj;
                ok = ac[i] == ac[i - 1] && (rr >> 31) == (rj >> 31) && dq > 0 && dq <= ANI_BP_BAND && dq <= ANI_MAX_LIN && dr == dq;
            }
            const unsigned long long m = __ballot(ok);
            if (lane == 0) lmask[b0 >> 6] = m;
        }
        __builtin_amdgcn_wave_barrier();
    }
    {
    int32_t fprev = 0, runmax = -0x40000000;
    for (uint32_t i = 0; i < n;) {
        if (LADDERS && fprev == runmax) {
            const uint32_t sh = i & 63u;
            const unsigned long long m = lmask[i >> 6] >> sh;
            if (m & 1ull) {
                // a stretch of marked anchors (up to the end of this block of 64): settled at once
                uint32_t L = (~m) ? (uint32_t)__ffsll((long long)~m) - 1u : 64u;
                L = L < 64u - sh ? L : 64u - sh;
                L = L < n - i ? L : n - i;
                if (lane < L) { f[i + lane] = fprev + ANI_ANCHOR_SCORE * (int32_t)(lane + 1u); bp[i + lane] = i + lane; }    // bp = predecessor index + 1
                fprev += ANI_ANCHOR_SCORE * (int32_t)L;
                runmax = fprev;
                i += L;
                SLOW_STAT(st_stretch);
                __builtin_amdgcn_wave_barrier();
                continue;
            }
        }
        const int32_t qpi = (int32_t)qp[i];
        const uint32_t rr = ar[i], rc = ac[i];
        const int32_t rp = (int32_t)(rr & 0x7FFFFFFFu);
        const uint32_t rev = rr >> 31;
        uint32_t key = 0;     // (score << 6) | (63 - lane): the maximum is the best score, nearest on ties
        if (lane < i && lane < ANI_BAND) {
            const uint32_t j = i - 1 - lane;
            const int32_t dq = qpi - (int32_t)qp[j];
            const uint32_t rj = ar[j];
            if (dq <= ANI_BP_BAND && ac[j] == rc && (rj >> 31) == rev) {
                const int32_t rpj = (int32_t)(rj & 0x7FFFFFFFu);
                const int32_t dr = rev ? rpj - rp : rp - rpj;
                if (dq > 0 && dr > 0 && dq <= ANI_MAX_LIN && dr <= ANI_MAX_LIN) {
                    const int32_t gap = dq > dr ? dq - dr : dr - dq;
                    if (gap <= ANI_MAX_GAP) {
                        const int32_t sc = f[j] + ANI_ANCHOR_SCORE - gap;
                        if (sc > ANI_ANCHOR_SCORE) key = ((uint32_t)sc << 6) | (63u - lane);
                    }
                }
            }
        }
        key = wave_max_u32(key);
        const int32_t fi = key ? (int32_t)(key >> 6) : ANI_ANCHOR_SCORE;
        if (lane == 0) { f[i] = fi; bp[i] = key ? i - (63u - (key & 63u)) : 0u; }   // bp = predecessor index + 1
        fprev = fi;
        runmax = fi > runmax ? fi : runmax;
        i++;
        SLOW_STAT(st_full);
        __builtin_amdgcn_wave_barrier();
    }
    }
    if (!LADDERS) {
    // 3. chains: best end first (ties: lowest index); back-track until the start or a used anchor
    for (;;) {
        uint32_t key = 0;     // (score << 10) | (1023 - index)
        for (uint32_t i = lane; i < n; i += 64) {
            const int32_t v = f[i];
            if (v > ANI_ANCHOR_SCORE) {
                const uint32_t k2 = ((uint32_t)v << 10) | (1023u - i);
                key = k2 > key ? k2 : key;
            }
        }
        key = wave_max_u32(key);
        if (!key) break;
        const uint32_t besti = 1023u - (key & 1023u);
        const int32_t bestv = (int32_t)(key >> 10);
        if (lane == 0) {
            uint32_t cnt = 0, rmin = 0xFFFFFFFFu, rmax = 0;
            int32_t cur = (int32_t)besti, first = (int32_t)besti;
            while (cur >= 0) {
                const uint32_t b = bp[cur];
                if (b & USED_BIT) break;
                cnt++;
                first = cur;
                const uint32_t rp = ar[cur] & 0x7FFFFFFFu;
                rmin = rp < rmin ? rp : rmin;
                rmax = rp > rmax ? rp : rmax;
                cur = (int32_t)(b & 0x7FFFFFFFu) - 1;
            }
            if (cnt < ANI_MIN_ANCHORS) {
                f[besti] = (int32_t)0x80000000;
            } else {
                cur = (int32_t)besti;
                while (cur >= 0) {
                    const uint32_t b = bp[cur];
                    if (b & USED_BIT) break;
                    bp[cur] = b | USED_BIT;
                    f[cur] = (int32_t)0x80000000;
                    cur = (int32_t)(b & 0x7FFFFFFFu) - 1;
                }
                const uint32_t slot = atomicAdd(&pair_nch[pi], 1u);
                if (slot < pd.c_cap) {
                    ChainRec cr;
                    cr.score = bestv; cr.n = cnt; cr.n_seeds = qi[besti] - qi[first] + 1;
                    cr.q0 = qp[first]; cr.q1 = qp[besti]; cr.r0 = rmin; cr.r1 = rmax; cr.chunk = c;
                    chains[pd.c_base + slot] = cr;
                } else {
                    atomicOr(&flags[0], 8u);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    } else {
    // 3. chains, ladder by ladder.  bp[k] == k: anchor k chains to the anchor before it.  Every anchor learns the bottom of
    // its ladder (packed above its predecessor: bp = bottom << 16 | predecessor + 1); ut[s], kept where the record tags were,
    // says how far ladder s is used: anchors [s, ut[s]) belong to chains already taken
    uint32_t *ut = ac;
    {
        uint32_t carry = 0;
        for (uint32_t b0 = 0; b0 < n; b0 += 64) {
            const uint32_t k = b0 + lane;
            const uint32_t b = k < n ? bp[k] : 0u;
            const unsigned long long lad = __ballot(k >= 1 && k < n && b == k);
            const unsigned long long z = ~lad & ((2ull << lane) - 1ull);                 // anchors of this block, up to k, that start a ladder
            const uint32_t bot = z ? b0 + 63u - (uint32_t)__clzll((long long)z) : carry;
            if (k < n) { bp[k] = (bot << 16) | b; ut[k] = 0u; }
            carry = ~lad ? b0 + 63u - (uint32_t)__clzll((long long)~lad) : carry;
        }
        __builtin_amdgcn_wave_barrier();
    }
    for (;;) {
        uint32_t key = 0;     // (score << 10) | (1023 - index), over the anchors no chain has taken
        for (uint32_t i = lane; i < n; i += 64) {
            const int32_t v = f[i];
            if (v > ANI_ANCHOR_SCORE && ut[bp[i] >> 16] <= i) {
                const uint32_t k2 = ((uint32_t)v << 10) | (1023u - i);
                key = k2 > key ? k2 : key;
            }
        }
        key = wave_max_u32(key);
        if (!key) break;
        const uint32_t besti = 1023u - (key & 1023u);
        const int32_t bestv = (int32_t)(key >> 10);
        SLOW_STAT(st_chains);
        if (lane == 0) {
            // the walk, twice: first counting (a chain needs three anchors), then taking
            uint32_t cnt = 0, rmin = 0xFFFFFFFFu, rmax = 0, first = besti;
            for (int take = 0; take < 2; take++) {
                uint32_t cur = besti;
                for (;;) {
                    const uint32_t w = bp[cur], s = w >> 16, u = ut[s];
                    SLOW_STAT(st_walk);
                    if (u > cur) break;                                   // this anchor belongs to an earlier chain
                    const uint32_t lo = u > s ? u : s;                    // the ladder from here down, as far as it is free
                    if (take) ut[s] = cur + 1u;
                    else {
                        cnt += cur - lo + 1u;
                        first = lo;
                        const uint32_t ra = ar[cur] & 0x7FFFFFFFu, rb = ar[lo] & 0x7FFFFFFFu;    // monotone along a ladder
                        const uint32_t mn = ra < rb ? ra : rb, mx = ra > rb ? ra : rb;
                        rmin = mn < rmin ? mn : rmin;
                        rmax = mx > rmax ? mx : rmax;
                    }
                    if (lo > s) break;                                    // met the used lower end
                    const uint32_t pb = bp[s] & 0xFFFFu;                  // predecessor of the ladder's bottom, + 1
                    if (!pb) break;
                    cur = pb - 1u;
                }
                if (cnt < ANI_MIN_ANCHORS) { f[besti] = (int32_t)0x80000000; break; }
            }
            if (cnt >= ANI_MIN_ANCHORS) {
                const uint32_t slot = atomicAdd(&pair_nch[pi], 1u);
                if (slot < pd.c_cap) {
                    ChainRec cr;
                    cr.score = bestv; cr.n = cnt; cr.n_seeds = qi[besti] - qi[first] + 1;
                    cr.q0 = qp[first]; cr.q1 = qp[besti]; cr.r0 = rmin; cr.r1 = rmax; cr.chunk = c;
                    chains[pd.c_base + slot] = cr;
                } else {
                    atomicOr(&flags[0], 8u);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    }
#ifdef SKDER_SLOW_STATS
    if (lane == 0) { atomicAdd(flags + 8, n); atomicAdd(flags + 9, st_full); atomicAdd(flags + 10, st_stretch); atomicAdd(flags + 11, st_chains); atomicAdd(flags + 12, st_walk); atomicAdd(flags + 13, 1u); }
#endif
    }   // declined chunks of this wave
}

// (num/den)^(1/15): Newton on doubles, + - * / only (ani_oracle.c oracle_root)
__device__ __forceinline__ double root_k(uint64_t num, uint64_t den)
{
    if (den == 0 || num == 0) return 0.0;
    if (num >= den) return 1.0;
    const double x = (double)num / (double)den;
    double y = 1.0;
    const double km1 = (double)(ANI_K - 1), kk = (double)ANI_K;
    for (int it = 0; it < ANI_ROOT_ITERS; it++) {
        double yp = 1.0;
#pragma unroll
        for (int i = 0; i < ANI_K - 1; i++) yp = yp * y;
        const double yn = (km1 * y + x / yp) / kk;
        if (yn == y) break;      // a fixed point: every further iteration returns the same value (the result is unchanged)
        y = yn;
    }
    return y;
}

// the two-estimate ANI model of include/skder_amd_spec.h (ani_oracle.c oracle_model_ani)
__device__ __forceinline__ double model_ani(double ani_cell, double ani_span)
{
    const double d = ANI_CAL_CELL * (100.0 * (1.0 - ani_cell)) + ANI_CAL_SPAN * (100.0 * (1.0 - ani_span));
    double a = 1.0 - d / 100.0;
    if (a < 0.0) a = 0.0;
    if (a > 1.0) a = 1.0;
    return a;
}

// is chain j ranked before chain i? (score desc, q0 asc, r0 asc, q1 asc) -- ani_oracle.c cmp_chain
__device__ __forceinline__ bool better(const int32_t *sc, const uint32_t *q0, const uint32_t *r0, const uint32_t *q1, uint32_t j, uint32_t i)
{
    if (sc[j] != sc[i]) return sc[j] > sc[i];
    if (q0[j] != q0[i]) return q0[j] < q0[i];
    if (r0[j] != r0[i]) return r0[j] < r0[i];
    if (q1[j] != q1[i]) return q1[j] < q1[i];
    // chains equal in every key (repeats can yield two chains with the same ends and score): the sort of the oracle
    // puts one of them first, and that one drops the other; which one cannot matter, so the array order decides
    return j < i;
}

// GLOBAL = false: the chain arrays of the pair live in dynamic LDS (`cap_arg` chains, sized per batch by the
// host), one workgroup per pair of the batch.  GLOBAL = true: the same code for the few pairs with more
// chains than LDS holds (repeat-rich genomes): arrays in a global workspace, workgroup b handles pair
// glist[b] with capacity gcap[b] at gws + goff[b].
template <bool GLOBAL>
__global__ __launch_bounds__(256) void finalize_kernel_t(SetView A, SetView B, const PairDesc *__restrict__ pairs,
                                                         const ChainRec *__restrict__ fast_chains, const uint32_t *__restrict__ chunk_state,
                                                         const ChainRec *__restrict__ chains, const uint32_t *__restrict__ pair_nch,
                                                         const uint32_t *__restrict__ pair_na, PairOut *__restrict__ out,
                                                         uint32_t *__restrict__ flags, uint32_t *__restrict__ chunk_mark, uint32_t cap_arg,
                                                         unsigned char *__restrict__ gws, const uint64_t *__restrict__ goff,
                                                         const uint32_t *__restrict__ glist, const uint32_t *__restrict__ gcap)
{
    // 8 word arrays + 1 byte array + 1 u16 array of `lds_cap` chains
    extern __shared__ __attribute__((aligned(16))) unsigned char fin_smem[];
    const uint32_t pidx = GLOBAL ? glist[blockIdx.x] : blockIdx.x;
    const uint32_t lds_cap = GLOBAL ? gcap[blockIdx.x] : cap_arg;
    unsigned char *const arrays = GLOBAL ? gws + goff[blockIdx.x] : fin_smem;
    int32_t *sc = reinterpret_cast<int32_t *>(arrays);
    uint32_t *q0 = reinterpret_cast<uint32_t *>(arrays) + lds_cap, *q1 = q0 + lds_cap, *r0 = q1 + lds_cap, *r1 = r0 + lds_cap;
    uint32_t *ckc = r1 + lds_cap, *na = ckc + lds_cap, *nsd = na + lds_cap;   // ckc: the chain's chunk inside the pair
    uint8_t *state = reinterpret_cast<uint8_t *>(nsd + lds_cap);   // 0 unknown, 1 kept, 2 dropped
    uint16_t *order = reinterpret_cast<uint16_t *>(state + lds_cap);  // chain indices grouped by bin
    __shared__ unsigned long long s_cells, s_seeds, s_anch, s_span;
    __shared__ uint32_t s_kept, s_unknown, s_n;

    const PairDesc pd = pairs[pidx];
    const uint32_t tid = threadIdx.x;
    if (tid == 0) { s_cells = 0; s_seeds = 0; s_anch = 0; s_span = 0; s_kept = 0; s_unknown = 0; s_n = 0; }
    // what the last lane standing needs at the very end is requested now (the workgroup holds its LDS until then)
    const uint64_t len_q = ((pd.flags & 2u) ? B : A).meta[pd.q].total_len, len_r = ((pd.flags & 4u) ? B : A).meta[pd.r].total_len;
    const uint32_t n_anchors_pair = pair_na[pidx];
    __syncthreads();
    // gather: chains of the fast path (per-chunk slots) and of the slow path (per-pair list)
    uint32_t nslow = pair_nch[pidx];
    if (nslow > pd.c_cap) nslow = pd.c_cap;
    auto put = [&](const ChainRec &c) {
        const uint32_t d = atomicAdd(&s_n, 1u);
        if (d < lds_cap) {
            sc[d] = c.score; q0[d] = c.q0; q1[d] = c.q1; r0[d] = c.r0; r1[d] = c.r1; ckc[d] = c.chunk;
            na[d] = c.n; nsd[d] = c.n_seeds;
            state[d] = 0;
        }
    };
    // one chunk per thread: its state, then all of its chains at once (a 3 Mb genome has 150 chunks: one trip)
    for (uint32_t ck = tid; ck < pd.n_chunks; ck += 256) {
        const uint32_t st = chunk_state[pd.chunk_base + ck];
        if (st == CHUNK_SLOW || st == 0u) continue;
        const ChainRec *fc = fast_chains + (uint64_t)(pd.chunk_base + ck) * FAST_SLOTS;
        ChainRec c[FAST_SLOTS];
#pragma unroll
        for (uint32_t k = 0; k < FAST_SLOTS; k++) if (k < st) c[k] = fc[k];
#pragma unroll
        for (uint32_t k = 0; k < FAST_SLOTS; k++) if (k < st) put(c[k]);
    }
    for (uint32_t i = tid; i < nslow; i += 256) put(chains[pd.c_base + i]);
    __syncthreads();
    uint32_t n = s_n;
    if (n > lds_cap) {
        // more chains than the arrays hold: report the number wanted (n_chains = ~0 marks the record) and
        // leave the pair to a launch with enough room
        if (tid == 0) {
            atomicOr(&flags[0], 16u);
            PairOut o;
            memset(&o, 0, sizeof o);
            o.n_chains = 0xFFFFFFFFu; o.n_chains_all = n;
            out[pidx] = o;
        }
        return;
    }
    __syncthreads();
    // Spatial binning on the other genome so that a chain is compared only with chains that can
    // overlap it: bins of width 2^shift >= the longest chain, chains filed under the bin of r0; a chain
    // in bin b can only overlap chains of bins b-1, b, b+1.  (Exact for any input: a very long chain
    // just makes the bins wide.)
    __shared__ uint32_t s_maxlen, s_maxr;
    __shared__ uint32_t bin_start[FIN_BINS + 2], bin_fill[FIN_BINS + 1];
    __shared__ uint32_t wsum[4];
    if (tid == 0) { s_maxlen = 0; s_maxr = 0; }
    __syncthreads();
    {
        uint32_t ml = 0, mr = 0;
        for (uint32_t i = tid; i < n; i += 256) { ml = max(ml, r1[i] - r0[i]); mr = max(mr, r1[i]); }
        if (ml) atomicMax(&s_maxlen, ml);
        if (mr) atomicMax(&s_maxr, mr);
    }
    __syncthreads();
    uint32_t shift = 1;
    while ((1u << shift) <= s_maxlen) shift++;
    while ((s_maxr >> shift) >= FIN_BINS) shift++;
    const uint32_t nb_used = (s_maxr >> shift) + 1u;     // bins that can hold a chain (a 3 Mb genome with 20 kb chunks: ~90 of 1024)
    for (uint32_t b = tid; b <= nb_used; b += 256) bin_fill[b] = 0;
    __syncthreads();
    for (uint32_t i = tid; i < n; i += 256) atomicAdd(&bin_fill[r0[i] >> shift], 1u);
    __syncthreads();
    {
        uint32_t running = 0;
        for (uint32_t base = 0; base < nb_used; base += 256) {
            const uint32_t v = base + tid < nb_used ? bin_fill[base + tid] : 0u;
            uint32_t total;
            const uint32_t ex = block_excl_scan_256(v, wsum, total);
            bin_start[base + tid] = running + ex;
            running += total;
        }
        __syncthreads();
        if (tid == 0) { bin_start[nb_used] = n; bin_start[nb_used + 1] = n; }
    }
    __syncthreads();
    for (uint32_t b = tid; b <= nb_used; b += 256) bin_fill[b] = 0;
    __syncthreads();
    for (uint32_t i = tid; i < n; i += 256) {
        const uint32_t b = r0[i] >> shift;
        order[bin_start[b] + atomicAdd(&bin_fill[b], 1u)] = (uint16_t)i;
    }
    __syncthreads();
    // a chain is dropped when ONE better kept chain on the same record covers more than half of
    // its span on the other genome.  Chains without any better overlapping chain are kept at once;
    // the rest resolve in rounds, each chain waiting for its better overlapping chains.
    for (;;) {
        uint32_t my_unknown = 0;
        for (uint32_t i = tid; i < n; i += 256) {
            if (state[i]) continue;
            const uint32_t li = r1[i] - r0[i];
            bool dropped = false, pending = false;
            const uint32_t b = r0[i] >> shift;
            const uint32_t k0 = bin_start[b ? b - 1 : 0], k1 = bin_start[b + 2];    // bin_start[nb_used], [nb_used + 1] = n
            for (uint32_t k = k0; k < k1; k++) {
                const uint32_t j = order[k];
                if (j == i) continue;     // chains lie inside one record and positions are genome-linear: overlap implies the same record
                const uint32_t lo = r0[i] > r0[j] ? r0[i] : r0[j];
                const uint32_t hi = r1[i] < r1[j] ? r1[i] : r1[j];
                if (hi <= lo) continue;
                if ((uint64_t)ANI_REF_OVERLAP_DEN * (hi - lo) <= (uint64_t)ANI_REF_OVERLAP_NUM * li) continue;
                if (!better(sc, q0, r0, q1, j, i)) continue;
                const uint8_t sj = state[j];
                if (sj == 1) { dropped = true; break; }
                if (sj == 0) pending = true;
            }
            if (dropped) state[i] = 2;
            else if (!pending) state[i] = 1;
            else my_unknown++;
        }
        if (my_unknown) atomicAdd(&s_unknown, my_unknown);
        __syncthreads();
        const uint32_t u = s_unknown;
        __syncthreads();
        if (tid == 0) s_unknown = 0;
        __syncthreads();
        if (!u) break;
    }
    // sums over the kept chains; the cells (chunks) that hold one are marked in global memory -- a pair
    // can have any number of chunks -- and their seed counts summed from the chunk table afterwards (an LDS bitmap for the
    // marks was measured: slower, 4.4 against 4.0 ms)
    uint32_t *mark = chunk_mark + pd.chunk_base;
    for (uint32_t i = tid; i < pd.n_chunks; i += 256) mark[i] = 0u;
    __syncthreads();
    unsigned long long sd = 0, an = 0, sp = 0, cs = 0;
    uint32_t kept = 0;
    for (uint32_t i = tid; i < n; i += 256) {
        if (state[i] != 1) continue;
        sd += nsd[i];
        an += na[i];
        sp += q1[i] - q0[i];
        kept++;
        mark[ckc[i]] = 1u;
    }
    __syncthreads();
    {
        const SetView &QS = (pd.flags & 2u) ? B : A;
        const uint32_t *cst = QS.chunk_start + pd.q_chunk_off;
        for (uint32_t i = tid; i < pd.n_chunks; i += 256)
            if (mark[i]) cs += cst[i + 1] - cst[i];
    }
    // wave-level reduction first: 4 LDS atomics per sum and workgroup instead of 256
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        cs += __shfl_down(cs, o, 64); sd += __shfl_down(sd, o, 64);
        an += __shfl_down(an, o, 64); sp += __shfl_down(sp, o, 64);
        kept += __shfl_down(kept, o, 64);
    }
    if ((tid & 63u) == 0 && (kept || cs)) {
        atomicAdd(&s_cells, cs); atomicAdd(&s_seeds, sd); atomicAdd(&s_anch, an); atomicAdd(&s_span, sp);
        atomicAdd(&s_kept, kept);
    }
    __syncthreads();
    if (tid >= 64) return;
    // the two 15-th roots are serial double arithmetic: one lane each
    double root = 0.0;
    if (tid < 2 && s_seeds) root = root_k(s_anch, tid == 0 ? s_cells : s_seeds);
    const double root_cell = __shfl(root, 0, 64), root_span = __shfl(root, 1, 64);
    if (tid == 0) {
        PairOut o;
        o.cell_seeds = s_cells; o.sum_seeds = s_seeds; o.sum_anchors = s_anch; o.sum_span = s_span;
        o.n_chains = s_kept; o.n_chains_all = n; o.n_anchors = n_anchors_pair; o.pad = 0;
        o.ani_raw = 0.0; o.ani_span = 0.0; o.ani = 0.0;
        if (s_seeds) {
            o.ani_raw = root_cell;
            o.ani_span = root_span;
            o.ani = model_ani(o.ani_raw, o.ani_span);
        }
        const double Bv = (double)(s_span + (unsigned long long)ANI_PAD * s_kept);
        const uint64_t tq = len_q, tr = len_r;
        double afq = tq ? Bv / (double)tq : 0.0, afr = tr ? Bv / (double)tr : 0.0;
        if (afq > 1.0) afq = 1.0;
        if (afr > 1.0) afr = 1.0;
        o.af_q = afq; o.af_r = afr;
        out[pidx] = o;
    }
}

// ---------------------------------------------------------------------------------------------
// host orchestration

static SetView view_of(skder_sketches *s)
{
    SetView v;
    v.meta = s->d_meta.p;
    v.pkmer = s->seed_kmer.p; v.pgpos = s->seed_gpos.p; v.pchunk = s->pchunk.p; v.pcs = s->pcs.p;
    v.skmer = s->skmer.p; v.sgpos = s->sgpos.p; v.sctg = s->sctg.p; v.stag = s->stag.p; v.boff = s->boff.p;
    v.chunk_start = s->chunk_start.p; v.rec_goff = s->d_rec_goff.p;
    return v;
}

// which genome is chunked: smaller T*(T/n_records); ties: fewer seeds, fewer markers, then the query
// (ani_oracle.c chunk_query)
static bool chunk_the_query(const GenomeMeta &ref, const GenomeMeta &query)
{
    double tq = (double)query.total_len, tr = (double)ref.total_len;
    double sq = tq * (tq / (double)(query.n_rec ? query.n_rec : 1));
    double sr = tr * (tr / (double)(ref.n_rec ? ref.n_rec : 1));
    if (sq != sr) return sq < sr;
    if (query.n_seeds != ref.n_seeds) return query.n_seeds < ref.n_seeds;     // content, not argument order
    if (query.n_markers != ref.n_markers) return query.n_markers < ref.n_markers;
    return true;
}

// work buffers of chain_pairs, kept across calls (grow-only) so that steady-state calls allocate nothing.
// CHAIN_SLOTS complete sets (slots): while the device works on the batches of two slots (one per queue), the host reads
// back and post-processes the third slot's results and prepares the descriptors of the next batch.
struct ChainSlot {
    DevBuf<PairDesc> d_pairs;
    DevBuf<uint32_t> chunk_state, chunk_mark, slow_list, counters, pair_na, pair_nch;
    DevBuf<uint32_t> hits, pair_nmulti, groups, over_list, flags;
    DevBuf<uint4> multi;
    DevBuf<RunRec> recs;
    DevBuf<uint32_t> pair_over, chunk_rec0, wg_pair, gen_list, gen_cnt, chunk_pair;
    std::vector<uint32_t> h_wg_pair;
    DevBuf<ChainRec> fast_chains, chains;
    DevBuf<PairOut> d_out;
    std::vector<PairDesc> hp;
    std::vector<JoinGroup> h_groups, h_groups2;
    std::vector<uint32_t> xcd_list[8];
    // pinned host mirrors of the small results
    PairOut *h_out = nullptr;
    size_t h_out_cap = 0;
    uint32_t *h_cnt = nullptr;      // [0..15] counters, [16] flags
    hipEvent_t ev[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // start, join, fast, slow, finalize, results on host; [6] run records
    size_t p0 = 0;
    uint32_t nb = 0, lds_cap = 0;
    uint64_t nchunks = 0, nrecs = 0;
    bool busy = false;
    hipStream_t st_join = nullptr, st = nullptr;     // the queues of the join and of the later stages (set per call)
    hipEvent_t ev_join = nullptr;                    // the join is done
    ~ChainSlot()
    {
        if (h_out) (void)hipHostFree(h_out);
        if (h_cnt) (void)hipHostFree(h_cnt);
        for (auto &e : ev) if (e) (void)hipEventDestroy(e);
        if (ev_join) (void)hipEventDestroy(ev_join);
    }
};
struct PairJob { uint32_t q, r, flags, orig; };
#define CHAIN_SLOTS 3
struct ChainWork {
    ChainSlot slot[CHAIN_SLOTS];
    std::vector<PairJob> jobs, jobs_sorted;
    std::vector<uint32_t> sort_start;
    // rare path (chunks with more anchors than the wave kernel holds in LDS)
    DevBuf<uint32_t> cap, abase, slow_n, a_qi, a_r, a_rctg, BP;
    DevBuf<int32_t> F;
    DevBuf<uint64_t> ORD;
    ScanWorkspace ws;
    hipEvent_t ev_order = nullptr;
    const uint32_t *oriented_ref = nullptr;      // chain_pairs_orient ran for this pair list (W.jobs holds the ordered work)
    size_t oriented_np = 0;
    hipStream_t stream3 = nullptr;
    ~ChainWork() { if (ev_order) (void)hipEventDestroy(ev_order); if (stream3) (void)hipStreamDestroy(stream3); }
};
static ChainWork *chain_work(skder_ctx *ctx)
{
    if (!ctx->chain_work) {
        ctx->chain_work = new ChainWork();
        ctx->chain_work_free = [](void *p) { delete static_cast<ChainWork *>(p); };
    }
    return static_cast<ChainWork *>(ctx->chain_work);
}

// orientation of every pair, then the work ordered by the probed genome (R): consecutive workgroups probe the same hash
// table, which keeps it in the XCD's L2.  Needs the genome table's lengths and counts only (there from index_begin on), so
// triangle_rows_impl / rectangle_impl call it while the index kernels are still running; chain_pairs does it itself otherwise.
static void chain_pairs_orient(skder_sketches *SA, skder_sketches *SB, const std::vector<uint32_t> &pref, const std::vector<uint32_t> &pquery)
{
    ChainWork &W = *chain_work(SA->ctx);
    const size_t np = pref.size();
    const auto t_host0 = std::chrono::steady_clock::now();
    std::vector<PairJob> &jobs = W.jobs, &sorted = W.jobs_sorted;     // kept across calls: no fresh pages to fault in
    jobs.resize(np); sorted.resize(np);
    {
        // orientation + stable counting sort by (set of R, R) on a few host threads: every thread owns a
        // contiguous range of pairs, counts its keys, and scatters into the slots that the prefix over
        // (key, thread) reserves for it -- the order inside a group stays the screen's
        const size_t na = SA->n_genomes, nbk = na + SB->n_genomes + 1;
        const unsigned T = np >= 65536 ? 4u : 1u;
        std::vector<uint32_t> &cnt = W.sort_start;
        cnt.assign((size_t)T * nbk, 0);
        auto keyof = [&](const PairJob &j) { return (size_t)((j.flags & 4u) ? na + j.r : j.r); };
        auto range = [&](unsigned t) { return std::make_pair(np * t / T, np * (t + 1) / T); };
        auto orient = [&](unsigned t) {
            const auto rg = range(t);
            uint32_t *c = cnt.data() + (size_t)t * nbk;
            for (size_t p = rg.first; p < rg.second; p++) {
                const GenomeMeta &mr = SA->h_meta[pref[p]], &mq = SB->h_meta[pquery[p]];
                const bool cq = chunk_the_query(mr, mq);
                PairJob j;
                j.q = cq ? pquery[p] : pref[p];
                j.r = cq ? pref[p] : pquery[p];
                j.flags = (cq ? 1u : 0u) | (cq ? 2u : 0u) | (cq ? 0u : 4u);   // Q in B iff cq; R in B iff !cq
                j.orig = (uint32_t)p;
                jobs[p] = j;
                c[keyof(j)]++;
            }
        };
        auto scatter = [&](unsigned t) {
            const auto rg = range(t);
            uint32_t *c = cnt.data() + (size_t)t * nbk;
            for (size_t p = rg.first; p < rg.second; p++) sorted[c[keyof(jobs[p])]++] = jobs[p];
        };
        auto run = [&](auto fn) {
            if (T == 1) { fn(0u); return; }
            std::vector<std::thread> th;
            for (unsigned t = 1; t < T; t++) th.emplace_back(fn, t);
            fn(0u);
            for (auto &x : th) x.join();
        };
        run(orient);
        uint32_t running = 0;
        for (size_t k = 0; k < nbk; k++)
            for (unsigned t = 0; t < T; t++) {
                uint32_t &c = cnt[(size_t)t * nbk + k];
                const uint32_t v = c;
                c = running;
                running += v;
            }
        run(scatter);
        jobs.swap(sorted);
    }
    W.oriented_ref = pref.data(); W.oriented_np = np;
    if (getenv("SKDER_AMD_DEBUG"))
        fprintf(stderr, "[skder_amd] host: orient+sort of %zu pairs %.2f ms\n", np,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_host0).count());
}

// pairs: (ref genome in set A, query genome in set B); for the triangle A == B.
void chain_pairs(skder_sketches *SA, skder_sketches *SB, const std::vector<uint32_t> &pref, const std::vector<uint32_t> &pquery,
                 std::vector<skder_edge_t> &edges)
{
    skder_ctx *ctx = SA->ctx;
    hipStream_t st = ctx->stream;
    const size_t np = pref.size();
    size_t budget = 6u << 20;   // chunks (work items) per batch
    int nqueues = 1;
    uint32_t join_group_max = 8;       // pairs per join workgroup (8: 24.6 ms per step of the benchmark; 16: 25.9; 4: 24.7; 32: 28.8)
    if (const char *e = getenv("SKDER_AMD_JOIN_GROUP")) join_group_max = (uint32_t)atoi(e);
    hipStream_t queues[3] = {st, st, st};
    if (const char *e = getenv("SKDER_AMD_CHUNK_BUDGET")) budget = strtoull(e, nullptr, 10);
    ChainWork &W = *chain_work(ctx);
    const SetView VA = view_of(SA), VB = view_of(SB);
    // debugging switches: SKDER_AMD_NO_XCD keeps the join's groups in launch order; SKDER_AMD_FORCE_SLOW sends every chunk down the slow path;
    // SKDER_AMD_NO_SIEVE hands every chunk with hits to chain_runs_kernel (to tell the two fast kernels apart behind a parity failure)
    int xcd_remap = (getenv("SKDER_AMD_NO_XCD") ? 0 : 1) | (getenv("SKDER_AMD_FORCE_SLOW") ? 2 : 0);
    if (const char *e = getenv("SKDER_AMD_NO_SIEVE")) xcd_remap |= atoi(e) ? 1024 : 0;
    double t_fast = 0, t_slow = 0, t_fin = 0, t_join = 0, t_runs = 0;
    uint64_t tot_anchors = 0, tot_slow = 0, tot_chunks = 0, tot_over = 0;
    // orientation + order by probed genome: done already if the caller used the wait for the index kernels for it
    if (!(W.oriented_ref == pref.data() && W.oriented_np == np)) chain_pairs_orient(SA, SB, pref, pquery);
    W.oriented_ref = nullptr; W.oriented_np = 0;
    std::vector<PairJob> &jobs = W.jobs;
    // ---- the chaining stage of a batch whose hit words exist: fast path, slow path, finalize, results
    auto chain_stage = [&](ChainSlot &S) {
        const uint32_t nb = S.nb;
        HIPCHECK(hipMemsetAsync(S.pair_nch.p, 0, nb * 4, S.st));
        HIPCHECK(hipMemsetAsync(S.pair_na.p, 0, nb * 4, S.st));
        HIPCHECK(hipMemsetAsync(S.counters.p, 0, 128, S.st));
        HIPCHECK(hipMemsetAsync(S.flags.p, 0, 64, S.st));
        HIPCHECK(hipEventRecord(S.ev[1], S.st));
        if (S.nchunks) {
            HIPCHECK(hipMemsetAsync(S.chunk_rec0.p, 0xFF, S.nchunks * 4, S.st));
            HIPCHECK(hipMemsetAsync(S.pair_over.p, 0, nb * 4, S.st));
            hipLaunchKernelGGL(run_extract_kernel, dim3(nb), dim3(256), 0, S.st, VA, VB, S.d_pairs.p, S.hits.p, S.recs.p, S.pair_over.p, S.chunk_rec0.p);
            HIPCHECK(hipEventRecord(S.ev[6], S.st));
            const unsigned nwg = (unsigned)((S.nchunks + 255) / 256);
            const uint32_t gen_cap = ((nwg + GEN_LISTS - 1u) / GEN_LISTS) * 256u;
            HIPCHECK(hipMemsetAsync(S.gen_cnt.p, 0, GEN_LISTS * 4, S.st));
            hipLaunchKernelGGL(chain_single_kernel, dim3(nwg), dim3(256), 0, S.st, VA, VB, S.d_pairs.p, nb, (uint32_t)S.nchunks, S.recs.p,
                               S.pair_over.p, S.chunk_rec0.p, S.wg_pair.p, S.multi.p, S.fast_chains.p, S.chunk_state.p, S.slow_list.p, S.counters.p,
                               S.gen_list.p, S.gen_cnt.p, gen_cap, S.pair_na.p, xcd_remap, S.chunk_pair.p);
            hipLaunchKernelGGL(chain_runs_kernel, dim3(nwg < 4096u ? nwg : 4096u), dim3(256), 0, S.st, VA, VB, S.d_pairs.p, nb, S.gen_list.p,
                               S.gen_cnt.p, gen_cap, S.recs.p, S.chunk_rec0.p, S.multi.p, S.fast_chains.p, S.chunk_state.p, S.slow_list.p, S.counters.p, S.pair_na.p,
                               S.chunk_pair.p);
        } else {
            HIPCHECK(hipEventRecord(S.ev[6], S.st));
        }
        HIPCHECK(hipEventRecord(S.ev[2], S.st));
        // declined chunks: one wavefront each, in LDS (count read on the device); the rare chunk with more
        // than 1024 anchors is put on over_list and dealt with after the batch's results are back
        if (S.nchunks) {
            const uint64_t want = (S.nchunks + SLOWW_WAVES - 1) / SLOWW_WAVES;
            static const bool ladders = getenv("SKDER_AMD_SLOW_PLAIN") == nullptr;      // (the per-anchor form, for A/B runs)
            hipLaunchKernelGGL(ladders ? slow_wave_kernel<true> : slow_wave_kernel<false>, dim3((unsigned)(want < 2048 ? want : 2048)), dim3(64 * SLOWW_WAVES), 0, S.st, VA, VB,
                               S.d_pairs.p, nb, S.slow_list.p, S.counters.p, S.hits.p, S.multi.p, S.chains.p, S.pair_nch.p, S.pair_na.p, S.over_list.p,
                               S.counters.p + 15, S.flags.p, S.chunk_pair.p);
        }
        HIPCHECK(hipEventRecord(S.ev[3], S.st));
        // LDS capacity of the finalize step: the most chains any pair of the batch can plausibly have
        // (1.5 per chunk + slack), rounded up; a batch in which some pair has more is finalized again with
        // the full 4096 (140 KB) when its results are read
        uint32_t max_chunks = 0;
        for (const PairDesc &d : S.hp) max_chunks = d.n_chunks > max_chunks ? d.n_chunks : max_chunks;
        uint32_t lds_cap = 512;
        while (lds_cap < FAST_SLOTS * max_chunks / 2 + 128 && lds_cap < 4096) lds_cap <<= 1;   // retried with more if a pair needs it
        S.lds_cap = lds_cap;
        if (nb)
            hipLaunchKernelGGL(finalize_kernel_t<false>, dim3(nb), dim3(256), lds_cap * 35u, S.st, VA, VB, S.d_pairs.p, S.fast_chains.p, S.chunk_state.p,
                               S.chains.p, S.pair_nch.p, S.pair_na.p, S.d_out.p, S.flags.p, S.chunk_mark.p, lds_cap, nullptr, nullptr, nullptr, nullptr);
        HIPCHECK(hipGetLastError());     // a rejected launch (resources) must not pass as an empty result
        HIPCHECK(hipEventRecord(S.ev[4], S.st));
        HIPCHECK(hipMemcpyAsync(S.h_out, S.d_out.p, nb * sizeof(PairOut), hipMemcpyDeviceToHost, S.st));
        HIPCHECK(hipMemcpyAsync(S.h_cnt, S.counters.p, 64, hipMemcpyDeviceToHost, S.st));
        HIPCHECK(hipMemcpyAsync(S.h_cnt + 16, S.flags.p, 4, hipMemcpyDeviceToHost, S.st));
#ifdef SKDER_SLOW_STATS
        HIPCHECK(hipMemcpyAsync(S.h_cnt + 24, S.flags.p + 8, 32, hipMemcpyDeviceToHost, S.st));
#endif
#ifdef SKDER_SIEVE_STATS
        HIPCHECK(hipMemcpyAsync(S.h_cnt + 24, S.counters.p + 24, 32, hipMemcpyDeviceToHost, S.st));
#endif
        HIPCHECK(hipEventRecord(S.ev[5], S.st));
    };
    // ---- one batch: descriptors (host), then everything on the stream without a host round trip
    uint32_t rec_div = 4;      // room for one run record per four seeds of the chunked genome (the 34 real C. granulosum genomes need one per six)
    if (const char *e = getenv("SKDER_AMD_REC_DIV")) rec_div = (uint32_t)atoi(e);
    auto enqueue = [&](ChainSlot &S, size_t p0, size_t batch_budget) -> size_t {
        std::vector<PairDesc> &hp = S.hp;
        hp.clear();
        uint64_t nchunks = 0, ccap = 0, nhits = 0, nmulti = 0, nrecs = 0;
        size_t p = p0;
        for (; p < np; p++) {
            const PairJob &jb = jobs[p];
            const GenomeMeta &Q = (jb.flags & 2u) ? SB->h_meta[jb.q] : SA->h_meta[jb.q];
            PairDesc d;
            memset(&d, 0, sizeof d);
            d.q = jb.q; d.r = jb.r; d.flags = jb.flags;
            {
                const GenomeMeta &R = (jb.flags & 4u) ? SB->h_meta[jb.r] : SA->h_meta[jb.r];
                if (Q.rep_cut != 0xFFFFFFFFu || R.total_len > (uint64_t)HIT_POS_MASK) d.flags |= 8u;
            }
            if (Q.chunk_off + Q.n_chunks + 1u > 0xFFFFFFFFull) throw SkError("chunk tables of the sketch set exceed 2^32 entries");
            d.q_chunk_off = (uint32_t)Q.chunk_off;
            d.n_chunks = Q.n_chunks;
            d.c_cap = 4u * Q.n_chunks + 64u;       // slow-path chains of the pair; made exact and retried when a pair needs more
            d.multi_cap = 256u + Q.n_seeds / 8u;
            if (d.multi_cap > 0x00FFFFF0u) d.multi_cap = 0x00FFFFF0u;
            if (!hp.empty() && nchunks + d.n_chunks > batch_budget) break;
            if (nchunks + d.n_chunks > 0x7FFF0000ull || ccap + d.c_cap > 0xFFFF0000ull || nhits + Q.n_seeds + 32u > 0xFFFF0000ull ||
                nmulti + d.multi_cap > 0xFFFF0000ull || nrecs + Q.n_seeds / rec_div + 64u > 0xFFFF0000ull) break;
            // hit words of the pair start at an entry congruent (mod 16) to the genome's seed offset:
            // chain_fast_kernel's two input streams then change their 64-byte line at the same seeds
            const uint64_t hb = nhits + ((Q.seed_off - nhits) & 15u);
            d.chunk_base = (uint32_t)nchunks; d.c_base = (uint32_t)ccap; d.hit_base = (uint32_t)hb; d.multi_base = (uint32_t)nmulti;
            d.seg_a = (uint32_t)(Q.seed_off & 3u);
            d.seg_per = ((Q.n_seeds + d.seg_a + SEG_SEEDS - 1u) / SEG_SEEDS + 3u) / 4u;
            if (!d.seg_per) d.seg_per = 1u;
            d.rec_base = (uint32_t)nrecs; d.rec_cap = (Q.n_seeds / rec_div + 64u) & ~3u;     // four quarters; more runs than a quarter holds: the pair takes the slow path
            nrecs += d.rec_cap;
            nchunks += d.n_chunks; ccap += d.c_cap; nhits = (hb + Q.n_seeds + 15u) & ~(uint64_t)15u; nmulti += d.multi_cap;
            hp.push_back(d);
        }
        const uint32_t nb = (uint32_t)hp.size();
        S.p0 = p0; S.nb = nb; S.nchunks = nchunks; S.nrecs = nrecs; S.busy = true;
        for (auto &e : S.ev) if (!e) HIPCHECK(hipEventCreate(&e));
        if (!S.ev_join) HIPCHECK(hipEventCreate(&S.ev_join));
        if (!S.h_cnt) HIPCHECK(hipHostMalloc(&S.h_cnt, 32 * sizeof(uint32_t)));
        if (S.h_out_cap < nb) {
            if (S.h_out) (void)hipHostFree(S.h_out);
            S.h_out = nullptr;
            S.h_out_cap = (size_t)nb + nb / 4 + 1024;
            HIPCHECK(hipHostMalloc(&S.h_out, S.h_out_cap * sizeof(PairOut)));
        }
        const auto t_al0 = std::chrono::steady_clock::now();
        // work buffers hold nothing worth keeping between batches: a buffer that has to grow is replaced, not copied
        auto grow = [&](auto &buf, size_t n) { buf.reserve(n, 0, S.st_join); buf.n = n; };
        grow(S.d_pairs, nb);
        grow(S.chunk_state, nchunks + 1); grow(S.chunk_mark, nchunks + 1); grow(S.slow_list, nchunks + 1); grow(S.over_list, nchunks + 1); grow(S.chunk_pair, nchunks + 1);
        grow(S.fast_chains, nchunks * FAST_SLOTS + 1);
        grow(S.counters, 32); grow(S.flags, 16);
        grow(S.pair_na, nb); grow(S.pair_nch, nb); grow(S.pair_nmulti, nb);
        grow(S.hits, nhits + 64); grow(S.multi, nmulti + 1);
        grow(S.recs, nrecs + 8); grow(S.pair_over, nb + 1); grow(S.chunk_rec0, nchunks + 1); grow(S.gen_list, nchunks + 256ull * GEN_LISTS + 1); grow(S.gen_cnt, GEN_LISTS);
        grow(S.chains, ccap + 1);
        grow(S.d_out, nb);
        if (getenv("SKDER_AMD_DEBUG"))
            fprintf(stderr, "[skder_amd] host: work buffers of the batch ready after %.2f ms\n",
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_al0).count());
        HIPCHECK(hipMemcpyAsync(S.d_pairs.p, hp.data(), nb * sizeof(PairDesc), hipMemcpyHostToDevice, S.st_join));
        {
            // first pair of every 256-chunk workgroup of the chaining kernel
            std::vector<uint32_t> &wp = S.h_wg_pair;
            const size_t nwg = (size_t)((nchunks + 255) / 256);
            wp.resize(nwg + 1);
            uint32_t pi = 0;
            for (size_t w = 0; w < nwg; w++) {
                const uint64_t t0 = (uint64_t)w * 256u;
                while (pi + 1u < nb && hp[pi + 1u].chunk_base <= t0) pi++;
                wp[w] = pi;
            }
            S.wg_pair.resize(nwg + 1, S.st_join);
            if (nwg) HIPCHECK(hipMemcpyAsync(S.wg_pair.p, wp.data(), nwg * 4, hipMemcpyHostToDevice, S.st_join));
        }
        HIPCHECK(hipMemsetAsync(S.pair_nmulti.p, 0, nb * 4, S.st_join));
        HIPCHECK(hipEventRecord(S.ev[0], S.st_join));
        {
            // groups of consecutive pairs that probe the same genome, at most 8 pairs each (load balance: a workgroup's ragged end)
            std::vector<JoinGroup> &hg = S.h_groups;
            hg.clear();
            for (uint32_t i = 0; i < nb;) {
                uint32_t j = i;
                while (j < nb && j - i < join_group_max && hp[j].r == hp[i].r && (hp[j].flags & 4u) == (hp[i].flags & 4u)) j++;
                JoinGroup g;
                g.pair_begin = i; g.pair_end = j;
                hg.push_back(g);
                i = j;
            }
            if ((xcd_remap & 1) && hg.size() >= 64) {
                // Workgroups are dealt round-robin to the 8 XCDs (workgroup b runs on XCD b mod 8).  The groups that probe
                // one genome -- consecutive in the list -- go to ONE XCD, so that they run there at about the same time
                // and share its L2 (the probed genome's positions are gathered from it) instead of filling all eight;
                // the genomes are dealt to the XCD with the least work so far (work: seeds streamed past), because an
                // XCD works through its own list and the kernel lasts as long as the slowest does -- equal eighths of
                // the list cost 60 % more join time on genomes of mixed sizes.  List ends are evened out at the end.
                std::vector<JoinGroup> &tmp = S.h_groups2;
                const size_t n = hg.size();
                tmp.resize(n);
                std::vector<uint32_t> (&lst)[8] = S.xcd_list;
                uint64_t load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (auto &l : lst) l.clear();
                for (size_t g0 = 0; g0 < n;) {
                    size_t g1 = g0;
                    uint64_t work = 0;
                    const PairDesc &first = hp[hg[g0].pair_begin];
                    while (g1 < n && hp[hg[g1].pair_begin].r == first.r && (hp[hg[g1].pair_begin].flags & 4u) == (first.flags & 4u)) {
                        for (uint32_t i = hg[g1].pair_begin; i < hg[g1].pair_end; i++)
                            work += ((hp[i].flags & 2u) ? SB->h_meta[hp[i].q] : SA->h_meta[hp[i].q]).n_seeds;
                        g1++;
                    }
                    int x = 0;
                    for (int y = 1; y < 8; y++) if (load[y] < load[x]) x = y;
                    load[x] += work;
                    for (size_t g = g0; g < g1; g++) lst[x].push_back((uint32_t)g);
                    g0 = g1;
                }
                size_t at[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (size_t bidx = 0; bidx < n; bidx++) {
                    int x = (int)(bidx % 8);
                    if (at[x] < lst[x].size()) { tmp[bidx] = hg[lst[x][at[x]++]]; continue; }
                    int y = 0;              // this XCD's list is used up: the last group of the list with the most left
                    for (int z = 1; z < 8; z++) if (lst[z].size() - at[z] > lst[y].size() - at[y]) y = z;
                    tmp[bidx] = hg[lst[y].back()];
                    lst[y].pop_back();
                }
                hg.swap(tmp);
            }
            S.groups.resize(hg.size() * 2, S.st_join);
            HIPCHECK(hipMemcpyAsync(S.groups.p, hg.data(), hg.size() * sizeof(JoinGroup), hipMemcpyHostToDevice, S.st_join));
            if (!ctx->chain_attr_set) {      // per context: the attribute belongs to the device, and a process may use several
                for (const void *k : {reinterpret_cast<const void *>(join_probe_kernel<0>), reinterpret_cast<const void *>(join_probe_kernel<1>)})
                    HIPCHECK(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, JOIN_SMEM_MAX + 64));
                HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(finalize_kernel_t<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             4096 * 35));
                ctx->chain_attr_set = true;
            }
            // LDS per workgroup: what the largest probed genome of the batch wants for a single pass; two
            // workgroups share a CU when that is at most half of it
            size_t want = 16384;
            for (uint32_t i = 0; i < nb; i++) {
                const GenomeMeta &R = (hp[i].flags & 4u) ? SB->h_meta[hp[i].r] : SA->h_meta[hp[i].r];
                const size_t w = join_need(R.bucket_bits, R.n_seeds);
                want = w > want ? w : want;
            }
            // SKDER_AMD_JOIN_V1: the per-entry probe of round 2 instead of the packed one (A/B; results identical)
            static const bool join_v1 = getenv("SKDER_AMD_JOIN_V1") != nullptr;
            auto join_kernel = join_v1 ? join_probe_kernel<0> : join_probe_kernel<1>;
            uint32_t join_smem = (uint32_t)(want < JOIN_SMEM_MAX ? want : JOIN_SMEM_MAX) / 64u * 64u + 64u;
            if (want <= JOIN_SMEM_TWO && join_smem > JOIN_SMEM_TWO) join_smem = JOIN_SMEM_TWO;
            if (getenv("SKDER_AMD_DEBUG")) {
                int per_cu = 0;
                (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(join_kernel), JOIN_THREADS, join_smem);
                fprintf(stderr, "[skder_amd] join: %zu workgroups of %u threads, %u bytes of LDS each: %d resident per CU (runtime's answer)\n", hg.size(), JOIN_THREADS, join_smem, per_cu);
            }
            hipLaunchKernelGGL(join_kernel, dim3((unsigned)hg.size()), dim3(JOIN_THREADS), join_smem, S.st_join, VA, VB, S.d_pairs.p,
                               reinterpret_cast<const JoinGroup *>(S.groups.p), S.hits.p, S.multi.p, S.pair_nmulti.p, join_smem);
            HIPCHECK(hipGetLastError());
        }
        // the later stages run on the slot's second queue, behind the join
        HIPCHECK(hipEventRecord(S.ev_join, S.st_join));
        if (S.st != S.st_join) HIPCHECK(hipStreamWaitEvent(S.st, S.ev_join, 0));
        chain_stage(S);
        return p;
    };
    auto check_flags = [](uint32_t h_flags) {
        if (h_flags & 4u) throw SkError("anchor buffer overflow in the slow path (internal error: counts and anchors disagree)");
        if (h_flags & 8u) throw SkError("chain buffer overflow after the capacities were made exact (internal error)");
        if (h_flags & 16u) throw SkError("finalize step: a pair still has more chains than its arrays hold (internal error)");
    };
    // ---- results of a batch: wait for its last copy, rare-path fix-up, edge records
    auto consume = [&](ChainSlot &S) {
        HIPCHECK(hipEventSynchronize(S.ev[5]));
        S.busy = false;
        const uint32_t nb = S.nb, nslow = S.h_cnt[0], nover = S.h_cnt[15];
        float ms;
        HIPCHECK(hipEventElapsedTime(&ms, S.ev[0], S.ev_join)); t_join += ms;
        HIPCHECK(hipEventElapsedTime(&ms, S.ev[1], S.ev[6])); t_runs += ms;
        HIPCHECK(hipEventElapsedTime(&ms, S.ev[6], S.ev[2])); t_fast += ms;
        HIPCHECK(hipEventElapsedTime(&ms, S.ev[2], S.ev[3])); t_slow += ms;
        HIPCHECK(hipEventElapsedTime(&ms, S.ev[3], S.ev[4])); t_fin += ms;
#ifdef SKDER_SIEVE_STATS
        fprintf(stderr, "[skder_amd] sieve: link %u, main-not-started-big %u, second-path %u, too-many-records %u, multi %u, third-stray %u, stray-near-main %u\n", S.h_cnt[24], S.h_cnt[25], S.h_cnt[26], S.h_cnt[28], S.h_cnt[29], S.h_cnt[30], S.h_cnt[31]);
#endif
#ifdef SKDER_RUNS_STATS
        { uint32_t x[8]; HIPCHECK(hipMemcpy(x, S.counters.p + 16, 32, hipMemcpyDeviceToHost));
          fprintf(stderr, "[skder_amd] run loop: %u chunks, %u wavefront rounds, %.1f lanes with a record per round, %.1f lanes not finished after it; %u rounds with a multi-occurrence seed (%u such lanes)\n",
                  S.h_cnt[11], S.h_cnt[12], S.h_cnt[13] / (double)(S.h_cnt[12] ? S.h_cnt[12] : 1), S.h_cnt[14] / (double)(S.h_cnt[12] ? S.h_cnt[12] : 1), x[1], x[2]); }
#endif
#ifdef SKDER_SLOW_STATS
        fprintf(stderr, "[skder_amd] slow path: %u chunks, %u anchors, %u full look-backs, %u stretches, %u chain ends, %u ladder visits\n", S.h_cnt[29], S.h_cnt[24], S.h_cnt[25], S.h_cnt[26], S.h_cnt[27], S.h_cnt[28]);
#endif
        if (getenv("SKDER_AMD_DEBUG")) {
            const uint32_t *hcnt = S.h_cnt;
            fprintf(stderr, "[skder_amd] batch: over %u; %u pairs %llu chunks, room for %llu run records, %u to the run loop, slow %u (none %u, slots %u, hits %u, ring %u, branch %u, score %u, qrep %u, inside %u, records-full %u, run-not-dominant %u)\n",
                    nover, nb, (unsigned long long)S.nchunks, (unsigned long long)S.nrecs, hcnt[11], nslow, hcnt[1], hcnt[2], hcnt[3], hcnt[4], hcnt[5], hcnt[6], hcnt[7], hcnt[8], hcnt[9], hcnt[10]);
        }
        // Rare-path fix-ups; the slot's buffers are untouched since (the batches in flight use the other slots).
        //  * nover: chunks the wave kernel could not hold go through the global-memory kernels;
        //  * flag 8: a pair produced more slow-path chains than its region holds (repeats: one chunk can
        //    chain to every copy).  pair_nch keeps counting past the capacity, so after a complete attempt it
        //    holds the number wanted: every pair gets exactly that and the chaining stage runs again;
        //  * flag 16: a pair has more chains than the LDS capacity chosen for finalize: again at 4096.
        auto finalize_and_fetch = [&]() {
            hipLaunchKernelGGL(finalize_kernel_t<false>, dim3(nb), dim3(256), S.lds_cap * 35u, S.st, VA, VB, S.d_pairs.p, S.fast_chains.p, S.chunk_state.p,
                               S.chains.p, S.pair_nch.p, S.pair_na.p, S.d_out.p, S.flags.p, S.chunk_mark.p, S.lds_cap, nullptr, nullptr, nullptr, nullptr);
            HIPCHECK(hipGetLastError());
            HIPCHECK(hipMemcpyAsync(S.h_out, S.d_out.p, nb * sizeof(PairOut), hipMemcpyDeviceToHost, S.st));
            HIPCHECK(hipMemcpyAsync(S.h_cnt + 16, S.flags.p, 4, hipMemcpyDeviceToHost, S.st));
            HIPCHECK(hipStreamSynchronize(S.st));
        };
        for (int attempt = 0;; attempt++) {
            uint32_t flags_seen = S.h_cnt[16];
            const uint32_t nover_now = S.h_cnt[15];
            if (nover_now) {
                W.cap.resize(nover_now + 1, S.st); W.abase.resize(nover_now + 1, S.st); W.slow_n.resize(nover_now + 1, S.st);
                HIPCHECK(hipMemsetAsync(S.flags.p, 0, 64, S.st));
                hipLaunchKernelGGL(slow_caps_kernel, dim3((nover_now + 4) / 4), dim3(256), 0, S.st, VA, VB, S.d_pairs.p, nb, S.over_list.p,
                                   nover_now, W.cap.p);
                exclusive_scan_u32(W.cap.p, W.abase.p, nover_now + 1, W.ws, S.st);
                uint32_t atotal = 0;
                HIPCHECK(hipMemcpyAsync(&atotal, W.abase.p + nover_now, 4, hipMemcpyDeviceToHost, S.st));
                HIPCHECK(hipStreamSynchronize(S.st));
                W.a_qi.resize(atotal + 1, S.st); W.a_r.resize(atotal + 1, S.st); W.a_rctg.resize(atotal + 1, S.st);
                W.F.resize(atotal + 1, S.st); W.BP.resize(atotal + 1, S.st); W.ORD.resize(atotal + 1, S.st);
                hipLaunchKernelGGL(slow_anchors_kernel, dim3((nover_now + 3) / 4), dim3(256), 0, S.st, VA, VB, S.d_pairs.p, nb, S.over_list.p,
                                   nover_now, W.abase.p, W.a_qi.p, W.a_r.p, W.a_rctg.p, W.slow_n.p, S.flags.p);
                hipLaunchKernelGGL(slow_chain_kernel, dim3((nover_now + 3) / 4), dim3(256), 0, S.st, VA, VB, S.d_pairs.p, nb, S.over_list.p,
                                   nover_now, W.abase.p, W.slow_n.p, W.a_qi.p, W.a_r.p, W.a_rctg.p, W.F.p, W.BP.p, W.ORD.p, S.chains.p, S.pair_nch.p,
                                   S.pair_na.p, S.flags.p);
                finalize_and_fetch();
                flags_seen = (flags_seen & 8u) | S.h_cnt[16];     // the wave kernel's overflow stays known
            }
            if ((flags_seen & 8u) && attempt == 0) {
                std::vector<uint32_t> want(nb);
                HIPCHECK(hipMemcpyAsync(want.data(), S.pair_nch.p, nb * 4ull, hipMemcpyDeviceToHost, S.st));
                HIPCHECK(hipStreamSynchronize(S.st));
                uint64_t ccap = 0;
                for (uint32_t i = 0; i < nb; i++) {
                    PairDesc &d = S.hp[i];
                    if (want[i] + 64u > d.c_cap) d.c_cap = want[i] + 64u;
                    d.c_base = (uint32_t)ccap;
                    ccap += d.c_cap;
                    if (ccap > 0xFFFF0000ull) throw SkError("chain buffer overflow: more than 2^32 slow-path chains in one batch of pairs");
                }
                S.chains.resize(ccap + 1, S.st);
                HIPCHECK(hipMemcpyAsync(S.d_pairs.p, S.hp.data(), nb * sizeof(PairDesc), hipMemcpyHostToDevice, S.st));
                chain_stage(S);
                HIPCHECK(hipEventSynchronize(S.ev[5]));
                continue;     // the over-list part runs again against the new regions
            }
            if ((flags_seen & 16u) && S.lds_cap < 4096) {
                S.lds_cap = 4096;
                HIPCHECK(hipMemsetAsync(S.flags.p, 0, 64, S.st));
                finalize_and_fetch();
                flags_seen = (flags_seen & ~16u) | S.h_cnt[16];
            }
            if (flags_seen & 16u) {
                // pairs with more than 4096 chains: the same step with its arrays in a global workspace
                std::vector<uint32_t> glist, gcap;
                std::vector<uint64_t> goff;
                uint64_t bytes = 0;
                for (uint32_t i = 0; i < nb; i++)
                    if (S.h_out[i].n_chains == 0xFFFFFFFFu) {
                        const uint32_t want = S.h_out[i].n_chains_all;
                        if (want > 65535u) throw SkError("pair with more than 65535 chains");   // chain indices are 16-bit in the overlap filter
                        const uint32_t capi = (want + 15u) & ~15u;      // keeps the byte and 16-bit arrays behind the word arrays aligned
                        glist.push_back(i); gcap.push_back(capi); goff.push_back(bytes);
                        bytes += ((uint64_t)capi * 35u + 255u) & ~(uint64_t)255u;
                    }
                if (!glist.empty()) {
                    DevBuf<unsigned char> gws;
                    DevBuf<uint32_t> d_list, d_cap;
                    DevBuf<uint64_t> d_off;
                    const size_t ng = glist.size();
                    gws.resize(bytes + 256, S.st); d_list.resize(ng, S.st); d_cap.resize(ng, S.st); d_off.resize(ng, S.st);
                    HIPCHECK(hipMemcpyAsync(d_list.p, glist.data(), ng * 4, hipMemcpyHostToDevice, S.st));
                    HIPCHECK(hipMemcpyAsync(d_cap.p, gcap.data(), ng * 4, hipMemcpyHostToDevice, S.st));
                    HIPCHECK(hipMemcpyAsync(d_off.p, goff.data(), ng * 8, hipMemcpyHostToDevice, S.st));
                    HIPCHECK(hipMemsetAsync(S.flags.p, 0, 64, S.st));
                    hipLaunchKernelGGL(finalize_kernel_t<true>, dim3((unsigned)ng), dim3(256), 0, S.st, VA, VB, S.d_pairs.p, S.fast_chains.p,
                                       S.chunk_state.p, S.chains.p, S.pair_nch.p, S.pair_na.p, S.d_out.p, S.flags.p, S.chunk_mark.p, 0u, gws.p,
                                       d_off.p, d_list.p, d_cap.p);
                    HIPCHECK(hipGetLastError());
                    HIPCHECK(hipMemcpyAsync(S.h_out, S.d_out.p, nb * sizeof(PairOut), hipMemcpyDeviceToHost, S.st));
                    HIPCHECK(hipMemcpyAsync(S.h_cnt + 16, S.flags.p, 4, hipMemcpyDeviceToHost, S.st));
                    HIPCHECK(hipStreamSynchronize(S.st));
                    flags_seen = (flags_seen & ~16u) | S.h_cnt[16];
                }
            }
            check_flags(flags_seen);
            tot_over += nover_now;
            break;
        }
        tot_slow += nslow; tot_chunks += S.nchunks;
        for (uint32_t i = 0; i < nb; i++) {
            const PairOut &o = S.h_out[i];
            tot_anchors += o.n_anchors;
            if (!o.n_chains || !(o.ani > 0.0)) continue;
            const bool cq = S.hp[i].flags & 1u;
            skder_edge_t e;
            e.ref = pref[jobs[S.p0 + i].orig]; e.query = pquery[jobs[S.p0 + i].orig];
            e.ani = o.ani;
            e.af_query = cq ? o.af_q : o.af_r;
            e.af_ref = cq ? o.af_r : o.af_q;
            e.n_chains = o.n_chains; e.n_anchors = o.n_anchors;
            e.aligned_bases = o.sum_span + (uint64_t)ANI_PAD * o.n_chains;
            e.sum_anchors = o.sum_anchors; e.sum_seeds = o.sum_seeds; e.cell_seeds = o.cell_seeds;
            e.ani_raw = o.ani_raw;
            edges.push_back(e);
        }
    };
    edges.reserve(edges.size() + np);
    // ---- several batches in flight, slot j on queue j: the stages of a batch are bound by different things (join: memory
    // latency at a third of the lanes' issue slots, long workgroups with a ragged end; run extraction: HBM; sieve and
    // finalize: dependent loads), so batches on different queues fill one another's gaps.  The first batches are
    // shorter (1/Q, 2/Q, ... of a batch), which puts the queues out of step.  SKDER_AMD_QUEUES=1: batch after batch
    // on the main queue.  (All joins on one queue and the later stages on a second one, behind their joins, measured no
    // gain at all: the join's two workgroups per CU take every wave slot and its next workgroup wins a freed one.)
    {
        uint64_t est = 0;
        for (size_t i = 0; i < np && est <= budget; i++) est += ((jobs[i].flags & 2u) ? SB->h_meta[jobs[i].q] : SA->h_meta[jobs[i].q]).n_chunks;
        nqueues = 2;              // three measured no faster than two (85.6 against 84.9 ms per step of the benchmark; one: 90.1)
        if (est <= budget / 4) nqueues = 1;          // a small job: one or two batches, nothing to overlap
        if (const char *e = getenv("SKDER_AMD_QUEUES")) nqueues = atoi(e);       // (tests force several queues on small jobs)
        if (nqueues < 1 || !ctx->stream2) nqueues = 1;
        if (nqueues > CHAIN_SLOTS) nqueues = CHAIN_SLOTS;
        if (nqueues > 2 && !W.stream3) HIPCHECK(hipStreamCreateWithFlags(&W.stream3, hipStreamNonBlocking));
        queues[0] = st; queues[1] = ctx->stream2; queues[2] = W.stream3;
        for (auto &S : W.slot) S.st_join = S.st = st;
        if (nqueues > 1) {      // what the caller queued on the main queue comes first
            if (!W.ev_order) HIPCHECK(hipEventCreateWithFlags(&W.ev_order, hipEventDisableTiming));
            HIPCHECK(hipEventRecord(W.ev_order, st));
            for (int j = 1; j < nqueues; j++) HIPCHECK(hipStreamWaitEvent(queues[j], W.ev_order, 0));
        }
    }
    const auto t_loop0 = std::chrono::steady_clock::now();
    try {
        size_t p0 = 0, k = 0;
        for (; p0 < np; k++) {
            ChainSlot &S = W.slot[k % CHAIN_SLOTS];
            if (S.busy) consume(S);            // batch k - CHAIN_SLOTS: results in batch order
            S.st_join = S.st = queues[k % (size_t)nqueues];
            p0 = enqueue(S, p0, k + 1 < (size_t)nqueues ? budget * (k + 1) / nqueues : budget);
        }
        for (size_t j = k < CHAIN_SLOTS ? 0 : k - CHAIN_SLOTS; j < k; j++)
            if (W.slot[j % CHAIN_SLOTS].busy) consume(W.slot[j % CHAIN_SLOTS]);
    } catch (...) {
        for (auto &S : W.slot) { (void)hipStreamSynchronize(S.st); (void)hipStreamSynchronize(S.st_join); S.busy = false; }
        throw;
    }
    if (getenv("SKDER_AMD_DEBUG"))
        fprintf(stderr, "[skder_amd] host: batch loop %.2f ms wall (kernels: join %.2f runs %.2f fast %.2f slow %.2f finalize %.2f)\n",
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_loop0).count(), t_join, t_runs, t_fast, t_slow, t_fin);
    // accumulated: a caller that works through its rows in blocks clears them once (chain_timing_reset)
    ctx->timing[3] += t_fast; ctx->timing[4] += t_slow; ctx->timing[5] += t_fin;
    ctx->timing[6] += (double)np; ctx->timing[7] += (double)tot_anchors;
    ctx->counters[0] += tot_chunks; ctx->counters[1] += tot_slow; ctx->counters[3] += tot_over;
    ctx->timing_join += t_join;
    ctx->timing_runs += t_runs;
}

static void chain_timing_reset(skder_ctx *ctx)
{
    ctx->timing[2] = ctx->timing[3] = ctx->timing[4] = ctx->timing[5] = ctx->timing[6] = ctx->timing[7] = 0.0;
    ctx->counters[0] = ctx->counters[1] = ctx->counters[3] = 0;
    ctx->timing_join = ctx->timing_runs = 0.0;
}

// Rows are screened and chained in BLOCKS whose worst case (every partner passes the screen) stays below
// 2^31 pairs: the screen counts and offsets pairs in 32 bits, and the host holds a few words per pair
// (92,700 genomes of one species screen to more than 2^32 pairs in one go).  5,000 genomes are one block.
static size_t rows_per_block(uint32_t n_partners)
{
    uint64_t budget = 1ull << 31;
    if (const char *e = getenv("SKDER_AMD_PAIR_BUDGET")) budget = strtoull(e, nullptr, 10);
    const uint64_t r = budget / (n_partners ? n_partners : 1u);
    return (size_t)(r ? r : 1u);
}

static void ensure_probed_indexed(skder_sketches *SA, skder_sketches *SB, const std::vector<uint32_t> &pr, const std::vector<uint32_t> &pq);

void triangle_rows_impl(skder_sketches *s, uint32_t row_begin, uint32_t row_stride, double screen_pct)
{
    skder_ctx *ctx = s->ctx;
    // a set that is not indexed yet: the seed index (stream2) is built while the marker screen runs -- the two
    // touch different arrays, and both are bound by latency rather than by any one unit of the GPU
    if (!s->indexed) index_begin(s, ctx->stream2);
    ctx->edges.clear();
    chain_timing_reset(ctx);
    std::vector<uint32_t> rows;
    for (uint32_t i = row_begin; i < s->n_genomes; i += (row_stride ? row_stride : 1)) rows.push_back(i);
    const size_t rpb = rows_per_block(s->n_genomes);
    std::vector<uint32_t> prow, ppart, sub;
    for (size_t b0 = 0; b0 < rows.size() || b0 == 0; b0 += rpb) {
        const size_t b1 = b0 + rpb < rows.size() ? b0 + rpb : rows.size();
        const std::vector<uint32_t> &blk = (b0 == 0 && b1 == rows.size()) ? rows : (sub.assign(rows.begin() + b0, rows.begin() + b1), sub);
        HIPCHECK(hipEventRecord(ctx->ev[9], ctx->stream));
        screen_pairs(s, s, blk, true, screen_pct, prow, ppart);
        HIPCHECK(hipEventRecord(ctx->ev[10], ctx->stream));
        HIPCHECK(hipStreamSynchronize(ctx->stream));
        float ms;
        HIPCHECK(hipEventElapsedTime(&ms, ctx->ev[9], ctx->ev[10]));
        ctx->timing[2] += ms;
        if (s->index_pending) chain_pairs_orient(s, s, prow, ppart);     // host work beside the index kernels
        index_impl(s);
        ensure_probed_indexed(s, s, prow, ppart);
        // triangle row (i, j): Ref = i, Query = j
        const auto t0 = std::chrono::steady_clock::now();
        chain_pairs(s, s, prow, ppart, ctx->edges);
        if (getenv("SKDER_AMD_DEBUG"))
            fprintf(stderr, "[skder_amd] host: rows %zu..%zu: screen %.2f ms (device), chain_pairs %.2f ms wall, %zu edges so far\n", b0, b1, ms,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), ctx->edges.size());
        if (b1 >= rows.size()) break;
    }
}

void rectangle_impl(skder_sketches *refs, skder_sketches *queries, double screen_pct)
{
    skder_ctx *ctx = refs->ctx;
    if (!queries->indexed) index_impl(queries);
    if (!refs->indexed) index_begin(refs, ctx->stream2);      // overlaps the screen, as in triangle_rows_impl
    ctx->edges.clear();
    chain_timing_reset(ctx);
    std::vector<uint32_t> rows(queries->n_genomes);
    for (uint32_t i = 0; i < queries->n_genomes; i++) rows[i] = i;
    const size_t rpb = rows_per_block(refs->n_genomes);
    std::vector<uint32_t> prow, ppart, sub;
    for (size_t b0 = 0; b0 < rows.size() || b0 == 0; b0 += rpb) {
        const size_t b1 = b0 + rpb < rows.size() ? b0 + rpb : rows.size();
        const std::vector<uint32_t> &blk = (b0 == 0 && b1 == rows.size()) ? rows : (sub.assign(rows.begin() + b0, rows.begin() + b1), sub);
        HIPCHECK(hipEventRecord(ctx->ev[9], ctx->stream));
        screen_pairs(refs, queries, blk, false, screen_pct, prow, ppart);
        HIPCHECK(hipEventRecord(ctx->ev[10], ctx->stream));
        HIPCHECK(hipStreamSynchronize(ctx->stream));
        float ms;
        HIPCHECK(hipEventElapsedTime(&ms, ctx->ev[9], ctx->ev[10]));
        ctx->timing[2] += ms;
        index_impl(refs);
        ensure_probed_indexed(refs, queries, ppart, prow);
        // rows are queries, partners are references
        chain_pairs(refs, queries, ppart, prow, ctx->edges);
        if (b1 >= rows.size()) break;
    }
}

// ---------------------------------------------------------------------------------------------
// pieces of triangle_rows / rectangle for callers that spread one pair matrix over several GPUs: candidate pairs of some
// rows (every GPU holds all markers), the genome each pair probes (its owner chains the pair), chaining of a pair list

void screen_rows_impl(skder_sketches *s, uint32_t row_begin, uint32_t row_stride, double screen_pct,
                      std::vector<uint32_t> &pref, std::vector<uint32_t> &pquery)
{
    skder_ctx *ctx = s->ctx;
    pref.clear(); pquery.clear();
    std::vector<uint32_t> rows;
    for (uint32_t i = row_begin; i < s->n_genomes; i += (row_stride ? row_stride : 1)) rows.push_back(i);
    const size_t rpb = rows_per_block(s->n_genomes);
    std::vector<uint32_t> a, b, sub;
    for (size_t b0 = 0; b0 < rows.size(); b0 += rpb) {
        const size_t b1 = b0 + rpb < rows.size() ? b0 + rpb : rows.size();
        sub.assign(rows.begin() + b0, rows.begin() + b1);
        screen_pairs(s, s, sub, true, screen_pct, a, b);
        pref.insert(pref.end(), a.begin(), a.end());
        pquery.insert(pquery.end(), b.begin(), b.end());
    }
    (void)ctx;
}

// the genome pair (ref, query) PROBES (the other one is cut into chunks): index inside the set
void pairs_probed_impl(skder_sketches *SA, skder_sketches *SB, const uint32_t *ref, const uint32_t *query, uint64_t n, uint32_t *probed,
                       uint8_t *probed_is_query)
{
    if (!SA->indexed || !SB->indexed) throw SkError("pairs_probed: index the sets first");
    for (uint64_t p = 0; p < n; p++) {
        if (ref[p] >= SA->n_genomes || query[p] >= SB->n_genomes) throw SkError("pairs_probed: genome index out of range");
        const bool cq = chunk_the_query(SA->h_meta[ref[p]], SB->h_meta[query[p]]);
        probed[p] = cq ? ref[p] : query[p];
        if (probed_is_query) probed_is_query[p] = cq ? 0 : 1;
    }
}

// The probed genome of every pair needs its bucket index here; a chunked genome whose own repetitive-k-mer filter is
// active needs it as well (slow chaining path).  A set indexed with skder_amd_sketches_index_part holds chunk tables only
// for the genomes another GPU owns: whatever a pair list needs beyond that is built first, so that no caller of the
// device-level interface (triangle_rows, rectangle, chain_pairs) can probe an index that was never written.
static void ensure_probed_indexed(skder_sketches *SA, skder_sketches *SB, const std::vector<uint32_t> &pr, const std::vector<uint32_t> &pq)
{
    if (SA->partial_index == 0 && SB->partial_index == 0) return;      // every genome has its bucket index: the usual case
    std::vector<uint32_t> needA, needB;
    for (size_t p = 0; p < pr.size(); p++) {
        const GenomeMeta &mr = SA->h_meta[pr[p]], &mq = SB->h_meta[pq[p]];
        const bool cq = chunk_the_query(mr, mq);
        if (cq) { if (!SA->full_index[pr[p]]) needA.push_back(pr[p]); if (mq.rep_cut != 0xFFFFFFFFu && !SB->full_index[pq[p]]) needB.push_back(pq[p]); }
        else { if (!SB->full_index[pq[p]]) needB.push_back(pq[p]); if (mr.rep_cut != 0xFFFFFFFFu && !SA->full_index[pr[p]]) needA.push_back(pr[p]); }
    }
    if (SA == SB) { needA.insert(needA.end(), needB.begin(), needB.end()); needB.clear(); }
    index_promote(SA, needA);
    if (SA != SB) index_promote(SB, needB);
}

void chain_pairs_impl(skder_sketches *SA, skder_sketches *SB, const uint32_t *ref, const uint32_t *query, uint64_t n)
{
    skder_ctx *ctx = SA->ctx;
    if (!SA->indexed || !SB->indexed) throw SkError("chain_pairs: index the sets first");
    for (uint64_t p = 0; p < n; p++)
        if (ref[p] >= SA->n_genomes || query[p] >= SB->n_genomes) throw SkError("chain_pairs: genome index out of range");
    ctx->edges.clear();
    chain_timing_reset(ctx);
    std::vector<uint32_t> pr(ref, ref + n), pq(query, query + n);
    ensure_probed_indexed(SA, SB, pr, pq);
    chain_pairs(SA, SB, pr, pq, ctx->edges);
}
