// chain_rows.hip -- the GENERAL chaining kernel: one 16-lane DPP row per (pair, 20 kb chunk), four chunks per wavefront.
//
// Unabridged restatement of oracle/ani_oracle.c oracle_pair() steps 3-4 for one chunk (banded DP
// f[i] = max(20, max_j f[j] + 20 - |dq - dr|), nearest first, strict '>'; chains best end first with back-tracking
// to the first used anchor), with every part spread over the 16 lanes of a row and all rows of a wavefront running the
// same instruction stream:
//   A. anchors   16 seeds per trip: hit word + position (coalesced), the 2..4-occurrence lists of the join, a bucket
//                walk only for seeds the join marked "too many"; a row scan (4 DPP adds) places them in LDS;
//   B. marks     16 anchors per trip: "continues the anchor in front of it on the same diagonal by a valid link" --
//                the stretch rule of DESIGN.md section 5 (at such an anchor whose predecessor holds the highest score so
//                far the look-back is settled without being run);
//   C. DP        a stretch of up to 32 marked anchors per step, then ONE look-back: 16 candidates per pass (one per
//                lane: two LDS reads, ~20 VALU), passes as long as the farthest candidate lies inside the 2500-base
//                band, a 4-instruction DPP row maximum of (score << 6 | nearness);
//   D. chains    best end = row maximum over the anchors, the walk ladder by ladder executed by
//                all 16 lanes alike (row-uniform values, broadcast LDS reads), the taken anchors voided 16 at a time.
// Nothing is "declined": a chunk leaves this kernel only when it does not fit (more than ROWS_MAXA anchors or 255
// seeds, a pair whose repetitive-k-mer filter is active or whose positions exceed 24 bits) -- then the one-wavefront-
// per-chunk kernel (384 anchors) and, behind that, the global-memory kernels take it.
//
// Why rows: the per-anchor work of the DP is a look-back over ~20 candidates (2500 bases at one seed per 125), so a
// 64-lane wavefront per chunk leaves two thirds of its lanes idle and pays ~60 instructions per anchor for ONE chunk;
// a lane per chunk (chain_runs_kernel) runs long divergent code at 11 of 64 lanes.  Sixteen lanes cover a look-back in
// one or two passes, four chunks share every instruction, and the control flow is row-uniform by construction.
#include "chain.h"

#define ROW_F_FAILED 1u        /* score field of an end whose chain had fewer than 3 anchors (0: taken by a chain) */
static_assert(ANI_BP_BAND <= ANI_MAX_LIN && ANI_BP_BAND + ANI_MAX_GAP <= ANI_MAX_LIN, "the 2500-base band implies both linear limits");
static_assert(ANI_BAND < 64, "nearness is packed into 6 bits");
static_assert(ROWS_MAXA <= 256 && ROWS_MAXA % 32 == 0 && ANI_ANCHOR_SCORE * ROWS_MAXA < 65536, "anchor indices are 8-bit, scores 16-bit");

struct __attribute__((aligned(16))) RowLds {
    uint32_t fq[ROWS_MAXA];     // score << 16 | position on the chunked genome, relative to the chunk's first seed
    uint32_t hw[ROWS_MAXA];     // hit word: position on the other genome | record tag << 24 | strand << 31
    uint16_t bp[ROWS_MAXA];     // predecessor + 1 (0: none)
    uint8_t bot[ROWS_MAXA];     // the bottom of the anchor's ladder (anchors whose predecessor is the anchor before)
    uint8_t qi[ROWS_MAXA];      // seed index inside the chunk
    uint8_t ut[ROWS_MAXA];      // per ladder bottom s: anchors [s, ut[s]) belong to chains already taken
    uint32_t marks[ROWS_MAXA / 32 + 2];
};

// orders the row's LDS traffic for the compiler (a wavefront's LDS operations execute in program order)
#define ROW_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

__device__ __forceinline__ uint32_t row_allmax(uint32_t v)       // maximum over the 16 lanes of a row, in every lane
{
    uint32_t t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x121, 0xF, 0xF, true); v = t > v ? t : v;   // row_ror:1
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x122, 0xF, 0xF, true); v = t > v ? t : v;   // row_ror:2
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xF, 0xF, true); v = t > v ? t : v;   // row_ror:4
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, true); v = t > v ? t : v;   // row_ror:8
    return v;
}
__device__ __forceinline__ uint32_t row_incl_scan(uint32_t v)    // inclusive prefix sum inside a row
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);
    return v;
}
__device__ __forceinline__ uint32_t row_last(uint32_t v)         // lane 15 of the row, in every lane (row_newbcast:15)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x15F, 0xF, 0xF, true);
}

__global__ __launch_bounds__(64 * ROWS_WAVES) void chain_rows_kernel(SetView A, SetView B, const PairDesc *__restrict__ pairs,
                                                                    const uint32_t *__restrict__ list, const uint32_t *__restrict__ n_ptr,
                                                                    const uint32_t *__restrict__ hits, const uint4 *__restrict__ multi,
                                                                    ChainRec *__restrict__ chains, uint32_t *__restrict__ pair_nch,
                                                                    uint32_t *__restrict__ pair_na, uint32_t *__restrict__ next_list,
                                                                    uint32_t *__restrict__ next_count, uint32_t *__restrict__ flags,
                                                                    const uint32_t *__restrict__ chunk_pair)
{
    __shared__ RowLds lds[ROWS_WAVES * 4];
    const uint32_t lane = threadIdx.x & 63u, rl = lane & 15u, row = lane >> 4, wv = threadIdx.x >> 6;
    RowLds &L = lds[wv * 4u + row];
    const uint32_t nlist = *n_ptr;
    const uint32_t stride = gridDim.x * ROWS_WAVES * 4u;
    for (uint32_t w = (blockIdx.x * ROWS_WAVES + wv) * 4u + row; w < nlist; w += stride) {
        ROW_SYNC();
        const uint32_t t = list[w];
        const uint32_t pi = chunk_pair[t];
        const PairDesc *pdp = pairs + pi;
        const uint32_t pflags = pdp->flags, pq = pdp->q, chunk_base = pdp->chunk_base;
        const SetView &QS = (pflags & 2u) ? B : A;
        const uint32_t c = t - chunk_base;
        const uint32_t *cst = QS.chunk_start + pdp->q_chunk_off + c;
        const uint32_t s0 = cst[0], s1 = cst[1];
        const uint64_t qoff = QS.meta[pq].seed_off;
        const uint32_t *qg = QS.pgpos + qoff;
        const uint32_t *hw_of = hits + pdp->hit_base;
        const uint32_t multi_base = pdp->multi_base;
        bool next = (pflags & 8u) != 0u || s1 - s0 > 255u;
        const uint32_t qbase = s1 > s0 ? qg[s0] : 0u;

        // ---- A. ordered anchors (several occurrences of one seed in ascending position on the other genome)
        uint32_t n = 0;
        for (uint32_t sb4 = s0; sb4 < s1 && !next; sb4 += 64u) {
            // 64 seeds per trip: the loads of four sub-trips are in flight together (the loop was the kernel's wait for memory)
            uint32_t hwv[4], qv[4];
#pragma unroll
            for (int u4 = 0; u4 < 4; u4++) {
                const uint32_t s = sb4 + 16u * (uint32_t)u4 + rl;
                hwv[u4] = s < s1 ? hw_of[s] : HIT_NONE;
                qv[u4] = s < s1 ? qg[s] : 0u;
            }
#pragma unroll
            for (int u4 = 0; u4 < 4; u4++) {
            const uint32_t sb = sb4 + 16u * (uint32_t)u4;
            if (sb >= s1 || next) break;
            const uint32_t s = sb + rl;
            uint32_t cnt = 0, first = 0, km = 0;
            const uint32_t qpos = qv[u4];
            uint4 mv = make_uint4(HIT_NONE, HIT_NONE, HIT_NONE, HIT_NONE);
            bool probe = false;
            {
                const uint32_t hw = hwv[u4];
                if (hw == HIT_MANY) probe = true;
                else if ((hw & 0xFF000000u) == HIT_MULTI) {
                    mv = multi[multi_base + (hw & 0x00FFFFFFu)];
                    cnt = 2u + (mv.z != HIT_NONE) + (mv.w != HIT_NONE);
                } else if (hw != HIT_NONE) { mv.x = hw; cnt = 1u; }
            }
            const uint32_t *rstag = nullptr;
            if (probe) {      // more than four occurrences (or no room in the join's lists): through the other genome's bucket index
                const SetView &RS = (pflags & 4u) ? B : A;
                const GenomeMeta *Rm = RS.meta + pdp->r;
                const uint64_t roff = Rm->seed_off;
                const uint32_t *rk = RS.skmer + roff, *rb = RS.boff + Rm->bucket_off;
                rstag = RS.stag + roff;
                km = (QS.pkmer + qoff)[s];
                const uint32_t kmer = km & SK_SEED_MASK;
                const uint32_t b = kmer_bucket(kmer, Rm->bucket_bits);
                const uint32_t lo = rb[b], hi = rb[b + 1];
                for (uint32_t e = lo; e < hi; e++) {
                    const uint32_t k2 = rk[e] & SK_SEED_MASK;
                    if (k2 == kmer) { if (!cnt) first = e; cnt++; }
                    else if (k2 > kmer) break;
                }
                if (cnt > Rm->rep_cut) cnt = 0;
            }
            const uint32_t incl = row_incl_scan(cnt);
            const uint32_t total = row_last(incl);
            if (n + total > ROWS_MAXA) { next = true; break; }      // row-uniform
            const uint32_t at = n + incl - cnt;
            if (cnt) {
                const uint32_t rel = qpos - qbase;
                if (probe) {
                    for (uint32_t u = 0; u < cnt; u++) {
                        const uint32_t idx = at + u;
                        L.fq[idx] = rel; L.hw[idx] = rstag[first + u] ^ (km & USED_BIT);
                        L.qi[idx] = (uint8_t)(s - s0); L.ut[idx] = 0;
                    }
                } else {
                    for (uint32_t u = 0; u < cnt; u++) {
                        const uint32_t idx = at + u, wd = u == 0 ? mv.x : (u == 1 ? mv.y : (u == 2 ? mv.z : mv.w));
                        L.fq[idx] = rel; L.hw[idx] = wd;
                        L.qi[idx] = (uint8_t)(s - s0); L.ut[idx] = 0;
                    }
                }
            }
            n += total;
            }
        }
        if (next) {       // does not fit a row: the one-wavefront-per-chunk kernel takes it
            if (rl == 0) next_list[atomicAdd(next_count, 1u)] = t;
            continue;
        }
        if (!n) continue;
        if (rl == 0) atomicAdd(&pair_na[pi], n);
#ifdef SKDER_ROWS_STATS
        uint32_t st_full = 0, st_pass = 0, st_stretch = 0, st_chains = 0, st_walk = 0;
#define ROWS_STAT(X) (X)++
#else
#define ROWS_STAT(X)
#endif
        if (rl < ROWS_MAXA / 32 + 2) L.marks[rl] = 0u;          // (one word more than the anchors fill: the 32-bit window behind the last anchor)
        ROW_SYNC();

        // ---- B. marks: anchor i continues anchor i - 1 by a valid link of gap 0
        for (uint32_t b0 = 0; b0 < n; b0 += 16u) {
            const uint32_t i = b0 + rl;
            bool ok = false;
            if (i >= 1u && i < n) {
                const uint32_t ai = L.fq[i], ap = L.fq[i - 1u], wi = L.hw[i], wp = L.hw[i - 1u];
                const uint32_t flip = (uint32_t)((int32_t)wi >> 31) & HIT_POS_MASK;
                ok = ((wi ^ wp) < (1u << HIT_POS_BITS)) && (ai - ap - 1u < (uint32_t)ANI_BP_BAND) && ((wi ^ flip) - ai == (wp ^ flip) - ap);
            }
            const unsigned long long m = __ballot(ok);
            if (rl == 0) reinterpret_cast<uint16_t *>(L.marks)[b0 >> 4] = (uint16_t)(m >> (row * 16u));
        }
        ROW_SYNC();

        // ---- C. banded DP
        {
            // The stretch rule: at an anchor that continues the anchor in front of it on the same diagonal
            // and whose predecessor holds the highest score so far (f[i-1] == runmax) the look-back is settled without being run
            // -- any candidate offers f[j] + 20 - gap <= runmax + 20, what the predecessor offers, and ties go to the nearest.
            // Its score is the new maximum, so the argument repeats along the stretch: up to 32 anchors per step.
            // (Measured and dropped: the nearest of the last FOUR anchors as the same-diagonal predecessor, so that a stray hit
            // inside the main path costs one look-back instead of two -- 7 % fewer look-backs, a heavier mark pass and one more
            // LDS read per step: 5.4 -> 6.4 ms on the real-structure set.)
            uint32_t i = 0, botprev = 0;
            int32_t fprev = 0, runmax = -1;
            while (i < n) {
                if (fprev == runmax) {
                    const uint32_t wd = i >> 5, sh = i & 31u;
                    const uint32_t m = __builtin_amdgcn_alignbit(L.marks[wd + 1u], L.marks[wd], sh);
                    const uint32_t len = m == 0xFFFFFFFFu ? 32u : (uint32_t)__ffs((int)~m) - 1u;      // marks behind the last anchor are 0
                    if (len) {
                        for (uint32_t k = rl; k < len; k += 16u) {
                            L.fq[i + k] |= (uint32_t)(fprev + ANI_ANCHOR_SCORE * (int32_t)(k + 1u)) << 16;
                            L.bp[i + k] = (uint16_t)(i + k);
                            L.bot[i + k] = (uint8_t)botprev;
                        }
                        i += len; fprev += ANI_ANCHOR_SCORE * (int32_t)len; runmax = fprev;
                        ROWS_STAT(st_stretch);
                        ROW_SYNC();
                    }
                }
                if (i < n)
                {
                    const uint32_t ai = L.fq[i], wi = L.hw[i];      // no score yet: ai is the position
                    const uint32_t flip = (uint32_t)((int32_t)wi >> 31) & HIT_POS_MASK;
                    const uint32_t yi = (wi ^ flip) - ai;
                    uint32_t key = 0;                            // score << 6 | 63 - distance: the maximum is the best score, nearest on ties
                    for (uint32_t d0 = 0; d0 < (uint32_t)ANI_BAND; d0 += 16u) {
                        const uint32_t dist = d0 + rl;           // candidate i - 1 - dist
                        uint32_t more = 0;
                        if (dist < i && dist < (uint32_t)ANI_BAND) {
                            const uint32_t j = i - 1u - dist;
                            const uint32_t aj = L.fq[j], wj = L.hw[j];
                            const uint32_t qj = aj & 0xFFFFu, dq = ai - qj;
                            more = dq <= (uint32_t)ANI_BP_BAND;
                            if (((wi ^ wj) < (1u << HIT_POS_BITS)) && dq - 1u < (uint32_t)ANI_BP_BAND) {
                                const int32_t dd = (int32_t)(yi - ((wj ^ flip) - qj));      // dr - dq
                                const int32_t ad = dd < 0 ? -dd : dd;
                                if (ad <= ANI_MAX_GAP && (int32_t)dq + dd > 0) {
                                    const int32_t sc = (int32_t)(aj >> 16) + ANI_ANCHOR_SCORE - ad;
                                    const uint32_t k2 = ((uint32_t)sc << 6) | (63u - dist);
                                    if (sc > ANI_ANCHOR_SCORE) key = k2 > key ? k2 : key;
                                }
                            }
                        }
                        ROWS_STAT(st_pass);
                        if (!row_last(more)) break;               // the farthest candidate of this pass is beyond the band (or the chunk's start)
                    }
                    key = row_allmax(key);
                    const uint32_t fi = key ? key >> 6 : (uint32_t)ANI_ANCHOR_SCORE;
                    const uint32_t pred1 = key ? i - (63u - (key & 63u)) : 0u;          // predecessor + 1
                    const uint32_t bot = (key && pred1 == i) ? botprev : i;
                    if (rl == 0) { L.fq[i] = ai | (fi << 16); L.bp[i] = (uint16_t)pred1; L.bot[i] = (uint8_t)bot; }
                    botprev = bot; fprev = (int32_t)fi; runmax = fprev > runmax ? fprev : runmax;
                    i++;
                    ROWS_STAT(st_full);
                    ROW_SYNC();
                }
            }
        }

        // ---- D. chains: best end first (ties: lowest index); back-track until the start or a used anchor
        const uint32_t c_base = pdp->c_base, c_cap = pdp->c_cap;
        for (;;) {
            uint32_t key = 0;     // score << 8 | 255 - index, over the anchors no chain has taken
            for (uint32_t k = rl; k < n; k += 16u) {
                const uint32_t v = L.fq[k] >> 16;
                const uint32_t k2 = (v << 8) | (255u - k);
                if (v > (uint32_t)ANI_ANCHOR_SCORE) key = k2 > key ? k2 : key;
            }
            key = row_allmax(key);
            if (!key) break;
            const uint32_t besti = 255u - (key & 255u), bestv = key >> 8;
            ROWS_STAT(st_chains);
            // the walk, twice (every lane of the row alike): first counting -- a chain needs three anchors --, then taking
            uint32_t cnt = 0, first = besti, rmin = 0xFFFFFFFFu, rmax = 0;
            {
                uint32_t cur = besti;
                for (;;) {
                    const uint32_t s = L.bot[cur], u = L.ut[s];
                    if (u > cur) break;                                   // this anchor belongs to an earlier chain
                    ROWS_STAT(st_walk);
                    const uint32_t lo = u > s ? u : s;                    // the ladder from here down, as far as it is free
                    cnt += cur - lo + 1u;
                    first = lo;
                    const uint32_t ra = L.hw[cur] & HIT_POS_MASK, rb = L.hw[lo] & HIT_POS_MASK;     // monotone along a ladder
                    const uint32_t mn = ra < rb ? ra : rb, mx = ra > rb ? ra : rb;
                    rmin = mn < rmin ? mn : rmin;
                    rmax = mx > rmax ? mx : rmax;
                    if (lo > s) break;                                    // met the used lower end
                    const uint32_t pb = L.bp[s];
                    if (!pb) break;
                    cur = pb - 1u;
                }
            }
            if (cnt < (uint32_t)ANI_MIN_ANCHORS) {
                if (rl == 0) L.fq[besti] = (L.fq[besti] & 0xFFFFu) | (ROW_F_FAILED << 16);
                ROW_SYNC();
                continue;
            }
            {
                uint32_t cur = besti;
                for (;;) {
                    const uint32_t s = L.bot[cur], u = L.ut[s];
                    if (u > cur) break;
                    const uint32_t lo = u > s ? u : s;
                    const uint32_t pb = L.bp[s];
                    for (uint32_t k = lo + rl; k <= cur; k += 16u) L.fq[k] &= 0xFFFFu;
                    if (rl == 0) L.ut[s] = (uint8_t)(cur + 1u);
                    if (lo > s || !pb) break;
                    cur = pb - 1u;
                }
            }
            if (rl == 0) {
                const uint32_t slot = atomicAdd(&pair_nch[pi], 1u);
                if (slot < c_cap) {
                    ChainRec cr;
                    cr.score = (int32_t)bestv; cr.n = cnt; cr.n_seeds = (uint32_t)L.qi[besti] - (uint32_t)L.qi[first] + 1u;
                    cr.q0 = qbase + (L.fq[first] & 0xFFFFu); cr.q1 = qbase + (L.fq[besti] & 0xFFFFu);
                    cr.r0 = rmin; cr.r1 = rmax; cr.chunk = c;
                    chains[c_base + slot] = cr;
                } else {
                    atomicOr(&flags[0], 8u);
                }
            }
            ROW_SYNC();
        }
#ifdef SKDER_ROWS_STATS
        if (rl == 0) { atomicAdd(flags + 8, n); atomicAdd(flags + 9, st_full); atomicAdd(flags + 10, st_stretch); atomicAdd(flags + 11, st_chains);
                       atomicAdd(flags + 12, st_walk); atomicAdd(flags + 13, 1u); atomicAdd(flags + 14, st_pass); }
#endif
    }
}

void launch_chain_rows(hipStream_t st, unsigned grid, SetView A, SetView B, const PairDesc *pairs, const uint32_t *list, const uint32_t *n_ptr,
                       const uint32_t *hits, const uint4 *multi, ChainRec *chains, uint32_t *pair_nch, uint32_t *pair_na, uint32_t *next_list,
                       uint32_t *next_count, uint32_t *flags, const uint32_t *chunk_pair)
{
    hipLaunchKernelGGL(chain_rows_kernel, dim3(grid), dim3(64 * ROWS_WAVES), 0, st, A, B, pairs, list, n_ptr, hits, multi, chains, pair_nch, pair_na,
                       next_list, next_count, flags, chunk_pair);
}
