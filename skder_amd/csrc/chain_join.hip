// chain_join.hip -- the JOIN of the chaining stage: hit words for every (pair, seed of the chunked genome)
#include <type_traits>

#include "chain.h"

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

void launch_join_probe(hipStream_t st, unsigned grid, uint32_t smem, SetView A, SetView B, const PairDesc *pairs, const JoinGroup *groups,
                       uint32_t *hits, uint4 *multi, uint32_t *pair_nmulti)
{
    hipLaunchKernelGGL(join_probe_kernel<1>, dim3(grid), dim3(JOIN_THREADS), smem, st, A, B, pairs, groups, hits, multi, pair_nmulti, smem);
}
void join_probe_allow_large_lds() { HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(join_probe_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, JOIN_SMEM_MAX + 64)); }
int join_probe_resident_per_cu(uint32_t smem)
{
    int per_cu = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(join_probe_kernel<1>), JOIN_THREADS, smem);
    return per_cu;
}
