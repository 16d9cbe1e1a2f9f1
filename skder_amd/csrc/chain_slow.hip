// chain_slow.hip -- the fall-through tiers behind the general kernel (chain_rows.hip): one WAVEFRONT per chunk in LDS
// (slow_wave_kernel, up to SLOWW_MAXA anchors; pairs whose repetitive-k-mer filter is active or whose positions exceed 24 bits),
// and the global-memory kernels for chunks beyond that (slow_caps / slow_anchors / slow_chain)
#include "chain.h"

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
    {
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
        if (fprev == runmax) {
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
    {
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

void launch_slow_wave(hipStream_t st, unsigned grid, SetView A, SetView B, const PairDesc *pairs, uint32_t npairs, const uint32_t *list,
                      const uint32_t *n_ptr, const uint32_t *hits, const uint4 *multi, ChainRec *chains, uint32_t *pair_nch, uint32_t *pair_na,
                      uint32_t *over_list, uint32_t *over_count, uint32_t *flags, const uint32_t *chunk_pair)
{
    hipLaunchKernelGGL(slow_wave_kernel, dim3(grid), dim3(64 * SLOWW_WAVES), 0, st, A, B, pairs, npairs, list, n_ptr, hits, multi, chains, pair_nch, pair_na,
                       over_list, over_count, flags, chunk_pair);
}
void launch_slow_caps(hipStream_t st, SetView A, SetView B, const PairDesc *pairs, uint32_t npairs, const uint32_t *list, uint32_t n, uint32_t *cap)
{
    hipLaunchKernelGGL(slow_caps_kernel, dim3((n + 4) / 4), dim3(256), 0, st, A, B, pairs, npairs, list, n, cap);
}
void launch_slow_anchors(hipStream_t st, SetView A, SetView B, const PairDesc *pairs, uint32_t npairs, const uint32_t *list, uint32_t n,
                         const uint32_t *abase, uint32_t *a_qi, uint32_t *a_r, uint32_t *a_rctg, uint32_t *slow_n, uint32_t *flags)
{
    hipLaunchKernelGGL(slow_anchors_kernel, dim3((n + 3) / 4), dim3(256), 0, st, A, B, pairs, npairs, list, n, abase, a_qi, a_r, a_rctg, slow_n, flags);
}
void launch_slow_chain(hipStream_t st, SetView A, SetView B, const PairDesc *pairs, uint32_t npairs, const uint32_t *list, uint32_t n,
                       const uint32_t *abase, const uint32_t *slow_n, const uint32_t *a_qi, const uint32_t *a_r, const uint32_t *a_rctg, int32_t *F,
                       uint32_t *BP, uint64_t *ORD, ChainRec *chains, uint32_t *pair_nch, uint32_t *pair_na, uint32_t *flags)
{
    hipLaunchKernelGGL(slow_chain_kernel, dim3((n + 3) / 4), dim3(256), 0, st, A, B, pairs, npairs, list, n, abase, slow_n, a_qi, a_r, a_rctg, F, BP, ORD,
                       chains, pair_nch, pair_na, flags);
}
