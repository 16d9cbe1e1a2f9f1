// chain_rruns.hip -- the RUN DP in row form: one 16-lane DPP row per (pair, 20 kb chunk), four chunks per wavefront, the chunk's
// RUNS held across the row's lanes.
//
// Same reasoning as the run loop (chain_runs.hip): a run record is a stretch of anchors each chained to the one
// before at a gap cost of at most RUN_GAP, so among a run's anchors only the last one can be the best predecessor of a later anchor,
// and a record's anchors behind its first follow at once when no other run can offer the second one more than the run itself does.
// What differs is where the state lives.  The run loop keeps the four most recent runs of its chunk in the registers of ONE lane and
// must give up whenever the answer depends on more than that: a look-back that reaches runs it no longer holds, an anchor with two
// successors, a path whose best end is not its last anchor, a fourth chain.  Here lane k of the row holds run k of the chunk -- all
// of them, up to sixteen:
//   look-back    every lane prices its run's last anchor as the predecessor of the current anchor (band, strand and record, gap, the
//                linear limits: the oracle's tests), one DPP row maximum of score << 13 | anchor ordinal << 1 | same-diagonal picks
//                the best offer, the nearest among equal ones -- the oracle's "nearest first, strict >";
//   runs         a same-diagonal winner grows in place; anything else opens the next lane with a link to the winner.  Links only
//                ever attach to the LAST anchor of a run (a run whose last anchor already has a successor is not grown: the anchor
//                opens a run of its own), so the anchors of a chunk form a forest over RUNS;
//   chains       the oracle's extraction -- ends by score (descending; ties: the earlier anchor), back-tracking to the start or to
//                an anchor an earlier chain took, fewer than three anchors: not a chain, the anchors stay free -- runs over that
//                forest exactly: inside a run the scores rise, so its last anchor is tried before any other of its anchors, takes
//                the whole run when it succeeds and leaves no chain to the others when it fails.  Branches, dead tails and any
//                number of chains cost nothing extra.
// Declined (to the general kernel, chain_rows.hip): a seed with more than four occurrences, more than sixteen runs, an anchor that
// an INTERIOR anchor of some run might precede (inside the run's extent, or within max_gap only by the run's diagonal steps), a
// record whose later anchors are not dominated by their own run, 4,000 anchors.
#include "chain.h"

#define RR_WAVES 2
#define RR_MAXRUNS 16u
#define RR_MAXANCH 4000u       /* anchor ordinals are packed into 12 bits */
static_assert(ANI_BAND < 4096 && RR_MAXANCH + 64u < 4096u, "anchor ordinals are packed into 12 bits");

__device__ __forceinline__ uint32_t rr_allmax(uint32_t v)        // maximum over the 16 lanes of a row, in every lane
{
    uint32_t t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x121, 0xF, 0xF, true); v = t > v ? t : v;   // row_ror:1
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x122, 0xF, 0xF, true); v = t > v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xF, 0xF, true); v = t > v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, true); v = t > v ? t : v;
    return v;
}
// the lanes of this lane's row for which p holds, as a 16-bit mask (one ballot for the wavefront, each row takes its quarter)
__device__ __forceinline__ uint32_t rr_rowmask(bool p, uint32_t rbase)
{
    return (uint32_t)(__ballot(p) >> rbase) & 0xFFFFu;
}
// v of another lane of the wavefront (the caller passes lanes of its own row: they are active whenever it is)
__device__ __forceinline__ uint32_t rr_from(uint32_t v, uint32_t src_lane)
{
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)v);
}

// items: either a flat list (list, n_ptr) or, when list == nullptr, the GEN_LISTS lists the sieve fills (gen_list, gen_cnt, gen_cap)
__global__ __launch_bounds__(64 * RR_WAVES) void chain_rruns_kernel(SetView A, SetView B, const PairDesc *__restrict__ pairs,
                                                                    const uint32_t *__restrict__ list, const uint32_t *__restrict__ n_ptr,
                                                                    const uint32_t *__restrict__ gen_list, const uint32_t *__restrict__ gen_cnt,
                                                                    uint32_t gen_cap, const RunRec *__restrict__ recs,
                                                                    const uint32_t *__restrict__ chunk_rec0, const uint4 *__restrict__ multi,
                                                                    ChainRec *__restrict__ fast_chains, uint32_t *__restrict__ chunk_state,
                                                                    ChainRec *__restrict__ chains, uint32_t *__restrict__ pair_nch,
                                                                    uint32_t *__restrict__ pair_na, uint32_t *__restrict__ next_list,
                                                                    uint32_t *__restrict__ next_count, uint32_t *__restrict__ stats,
                                                                    uint32_t *__restrict__ flags, const uint32_t *__restrict__ chunk_pair)
{
    __shared__ uint32_t g_off[GEN_LISTS + 1];
    uint32_t n_items;
    if (list) n_items = *n_ptr;
    else {
        // offsets of the GEN_LISTS lists laid end to end (a serial scan by one lane: 256 additions once per workgroup)
        if (threadIdx.x == 0) {
            uint32_t acc = 0;
            for (uint32_t k = 0; k < GEN_LISTS; k++) { g_off[k] = acc; acc += gen_cnt[k]; }
            g_off[GEN_LISTS] = acc;
        }
        __syncthreads();
        n_items = g_off[GEN_LISTS];
    }
    const uint32_t lane = threadIdx.x & 63u, rl = lane & 15u, rbase = lane & 48u, wv = threadIdx.x >> 6;
    const uint32_t stride = gridDim.x * RR_WAVES * 4u;
    const int32_t NEG = -0x40000000;
    for (uint32_t w0 = (blockIdx.x * RR_WAVES + wv) * 4u; w0 < n_items; w0 += stride) {
        const uint32_t w = w0 + (rbase >> 4);
        const bool live = w < n_items;
        uint32_t t = 0;
        if (live) {
            if (list) t = list[w];
            else {
                uint32_t lo = 0, hi = GEN_LISTS;
                while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (g_off[mid] <= w) lo = mid; else hi = mid; }
                t = gen_list[(uint64_t)lo * gen_cap + (w - g_off[lo])];
            }
        }
        const uint32_t pi = live ? chunk_pair[t] : 0u;
        const PairDesc pd = pairs[pi];
        const uint32_t idx0 = live ? chunk_rec0[t] : 0xFFFFFFFFu;
        const uint32_t c = t - pd.chunk_base;
        const SetView &QS = (pd.flags & 2u) ? B : A;
        const uint32_t s0 = live ? QS.chunk_start[pd.q_chunk_off + c] : 0u, s1 = live ? QS.chunk_start[pd.q_chunk_off + c + 1] : 0u;
        bool cplx = false;
        uint32_t cause = 0u;

        // this lane's run (lane rl < nruns)
        uint32_t q_last = 0, rr_last = 0, idx_last = 0, cnt = 0, pred = 0, qi_first = 0, q_first = 0, qi_last = 0, r_first = 0;
        int32_t f = 0, gs = 0;
        bool succ = false;                     // the run's last anchor is the predecessor of another run's first
        // the chunk (the same in all lanes of the row)
        uint32_t nruns = 0, ia = 0;
        int32_t runmax = NEG;

        const uint4 *prec = reinterpret_cast<const uint4 *>(recs + pd.rec_base);
        uint32_t idx = idx0;
        bool done = !live || idx == 0xFFFFFFFFu || s1 <= s0;
        uint4 a0 = make_uint4(0, 0, 0, 0), a1 = a0, b0 = a0, b1 = a0;
        if (!done) { a0 = prec[2u * idx]; a1 = prec[2u * idx + 1u]; b0 = prec[2u * idx + 2u]; b1 = prec[2u * idx + 3u]; }
        struct { uint32_t qi, q0, hw, q1, qi1, hw1, n, gsum; } rc;
        rc.qi = 0; rc.q0 = 0; rc.hw = HIT_NONE; rc.hw1 = HIT_NONE; rc.q1 = 0; rc.qi1 = 0; rc.n = 0; rc.gsum = 0;
        uint32_t pend = 0, g0 = HIT_NONE, g1 = HIT_NONE, g2 = HIT_NONE, g3 = HIT_NONE;
        // ---- the DP: one anchor per step (a record's first anchor, or one occurrence of a seed with 2..4 occurrences)
        for (;;) {
            bool have = pend != 0u;
            if (!done && !have) {
                if (a0.x == REC_LINK && s1 <= a0.z) done = true;       // the chunk ends with its quarter
                else if (a0.x == REC_LINK) {                            // the chunk goes on in the next quarter of the region
                    idx = a0.y;
                    a0 = prec[2u * idx]; a1 = prec[2u * idx + 1u];
                    if (a0.x < REC_LINK) { b0 = prec[2u * idx + 2u]; b1 = prec[2u * idx + 3u]; }
                } else if (a0.x >= s1) done = true;                     // records are in seed order (terminator: ~0)
                else {
                    have = true;
                    rc.qi = a0.x; rc.q0 = a0.y; rc.hw = a0.z;
                    rc.n = b0.w - a0.w; rc.gsum = b1.w - a1.w;           // running counts: this run's share
                    rc.q1 = b1.x; rc.hw1 = b1.y; rc.qi1 = b1.z;          // the hit in front of the next record ends this run
                    a0 = b0; a1 = b1;
                    idx++;
                    if (a0.x < REC_LINK) { b0 = prec[2u * idx + 2u]; b1 = prec[2u * idx + 3u]; }
                }
            }
            if (have) do {
                const uint32_t s = rc.qi;
                const int32_t qp = (int32_t)rc.q0;
                const uint32_t hw = rc.hw;
                if (pend == 0u) {            // a new record
                    if (hw == HIT_MANY) { cplx = true; cause = 2; break; }
                    if (ia + rc.n >= RR_MAXANCH) { cplx = true; cause = 6; break; }
                    pend = 1u; g0 = hw; g1 = g2 = g3 = HIT_NONE;
                    if ((hw & 0xFF000000u) == HIT_MULTI) {   // 2..4 occurrences, ascending gpos
                        const uint4 mv = multi[pd.multi_base + (hw & 0x00FFFFFFu)];
                        g0 = mv.x; g1 = mv.y; g2 = mv.z; g3 = mv.w;
                        pend = 2u + (g2 != HIT_NONE) + (g3 != HIT_NONE);
                    }
                }
                const uint32_t rr = g0;
                g0 = g1; g1 = g2; g2 = g3;
                pend--;
                const int32_t rp = (int32_t)(rr & HIT_POS_MASK);
                const uint32_t rev = rr >> 31;
                const uint32_t key = rr & HIT_KEY_MASK;     // strand + record tag
                const int32_t dg = rev ? rp + qp : rp - qp;
                // ---- look-back: this lane's run as the predecessor
                const bool has = rl < nruns;
                bool cand = false, f7 = false;
                int32_t sc = 0;
                uint32_t same_diag = 0;
                {
                    const int32_t dq = qp - (int32_t)q_last;
                    if (has && ia - idx_last <= (uint32_t)ANI_BAND && dq <= ANI_BP_BAND && (rr_last & HIT_KEY_MASK) == key) {
                        const int32_t rpj = (int32_t)(rr_last & HIT_POS_MASK);
                        const int32_t dr = rev ? rpj - rp : rp - rpj;
                        const int32_t ed = rev ? rpj + (int32_t)q_last : rpj - (int32_t)q_last;      // the run's diagonal at its last anchor
                        const int32_t off = dg > ed ? dg - ed : ed - dg;
                        if (off > ANI_MAX_GAP) {
                            if (off - gs <= ANI_MAX_GAP) f7 = true;      // an earlier anchor of a run with steps may be in reach where the last one is not
                        } else {
                            const int32_t rf = (int32_t)r_first;
                            const bool inside = rev ? (rp < rf && dr <= 0) : (rp > rf && dr <= 0);
                            if (dq <= 0 || inside) f7 = true;            // an interior anchor could be a valid predecessor where the last one is not
                            else if (dr > 0 && dq <= ANI_MAX_LIN && dr <= ANI_MAX_LIN) {
                                sc = f + ANI_ANCHOR_SCORE - off;
                                cand = sc > ANI_ANCHOR_SCORE;
                                same_diag = off == 0 ? 1u : 0u;
                            }
                        }
                    }
                }
                // the oracle's loop stops, nearest first, once its best offer reaches runmax + 20 (nothing older can beat or tie-break it):
                // a doubt about a run OLDER than such an offer is no doubt
                const uint32_t nq = rr_allmax((cand && sc >= runmax + ANI_ANCHOR_SCORE) ? idx_last + 1u : 0u);
                if (f7 && nq > idx_last + 1u) f7 = false;
                if (rr_rowmask(f7, rbase)) { cplx = true; cause = 7; break; }
                const uint32_t mykey = cand ? (((uint32_t)sc << 13) | (idx_last << 1) | same_diag) : 0u;
                const uint32_t mx = rr_allmax(mykey);
                const int32_t best = mx ? (int32_t)(mx >> 13) : ANI_ANCHOR_SCORE;
                const bool iwin = mx != 0u && mykey == mx;
                const uint32_t wmask = rr_rowmask(iwin, rbase);
                const uint32_t bj = wmask ? (uint32_t)__ffs((int)wmask) - 1u : 0u;
                bool grow = mx != 0u && (mx & 1u);
                if (grow && rr_rowmask(iwin && succ, rbase)) grow = false;      // its last anchor has a successor already: a run of its own
                uint32_t cur;
                if (grow) {
                    cur = bj;
                    if (iwin) { f = best; q_last = (uint32_t)qp; rr_last = rr; cnt += 1u; qi_last = s; idx_last = ia; }
                } else {
                    if (nruns >= RR_MAXRUNS) { cplx = true; cause = 3; break; }
                    cur = nruns;
                    if (iwin) succ = true;
                    if (rl == nruns) {
                        f = best; q_last = (uint32_t)qp; rr_last = rr; idx_last = ia; gs = 0; r_first = (uint32_t)rp; cnt = 1u;
                        pred = mx ? bj + 1u : 0u; qi_first = s; q_first = (uint32_t)qp; qi_last = s; succ = false;
                    }
                    nruns++;
                }
                ia++;
                runmax = best > runmax ? best : runmax;
                if (pend == 0u && rc.n > 1u) {
                    // ---- the run's other anchors: extensions of lane cur along the run, provided no other run can offer its second
                    // anchor more than the run itself does (chain_runs.hip has the argument): every other run is of another record or
                    // strand, beyond the 2500-base band already at the first anchor, further off than max_gap plus all the diagonal
                    // steps of both runs, or scores no more than this run + its diagonal distance (- 20 when the run has steps)
                    const uint32_t k0 = hw & HIT_KEY_MASK;
                    const int32_t rp0 = (int32_t)(hw & HIT_POS_MASK);
                    const int32_t d0 = (hw >> 31) ? rp0 + qp : rp0 - qp;
                    const int32_t G = (int32_t)rc.gsum, slack = G ? 2 * RUN_GAP : 0;
                    const int32_t fcur = (int32_t)rr_from((uint32_t)f, rbase + cur);
                    const int32_t f0 = fcur - slack;
                    bool harm = false;
                    if (rl < nruns && rl != cur && (rr_last & HIT_KEY_MASK) == k0 && qp - (int32_t)q_last <= ANI_BP_BAND) {
                        const int32_t de = (rr_last >> 31) ? (int32_t)(rr_last & HIT_POS_MASK) + (int32_t)q_last : (int32_t)(rr_last & HIT_POS_MASK) - (int32_t)q_last;
                        const int32_t doff = de > d0 ? de - d0 : d0 - de;
                        harm = !(doff - G - gs > ANI_MAX_GAP || f - doff <= f0);
                    }
                    if (rr_rowmask(harm, rbase)) { cplx = true; cause = 9; break; }
                    const uint32_t ext = rc.n - 1u;
                    if (rl == cur) {
                        q_last = rc.q1; rr_last = rc.hw1;
                        f = f + ANI_ANCHOR_SCORE * (int32_t)ext - G;
                        cnt += ext; idx_last = ia + ext - 1u; qi_last = rc.qi1; gs += G;
                    }
                    const int32_t fe = fcur + ANI_ANCHOR_SCORE * (int32_t)ext - G;
                    runmax = fe > runmax ? fe : runmax;
                    ia += ext;
                }
            } while (0);
            if (cplx) { done = true; pend = 0u; }
            if (!__any(!done || pend != 0u)) break;     // all four chunks of the wavefront are through
        }

        // ---- chains: ends by (score descending, anchor ordinal ascending) over the runs; a walk follows the links until the start or a
        // run an earlier chain took; fewer than ANI_MIN_ANCHORS anchors: no chain, the runs stay free (and that end is not tried again)
        uint32_t used = 0u, tried = 0u, nfin = 0u;
        bool ex = live && !cplx && nruns != 0u;
        ChainRec *slots = fast_chains + (uint64_t)t * FAST_SLOTS;
        for (;;) {
            const uint32_t k = (ex && rl < nruns && !used && !tried) ? (((uint32_t)f << 12) | (4095u - idx_last)) : 0u;
            const uint32_t mx = rr_allmax(k);
            if (!mx) ex = false;
            if (!__any(ex)) break;
            if (ex) {
                const uint32_t wl = (uint32_t)__ffs((int)rr_rowmask(k == mx, rbase)) - 1u;
                uint32_t n = 0, cur = wl, first = wl;
                for (;;) {
                    if (rr_from(used, rbase + cur)) break;
                    n += rr_from(cnt, rbase + cur);
                    first = cur;
                    const uint32_t p = rr_from(pred, rbase + cur);
                    if (!p) break;
                    cur = p - 1u;
                }
                if (n < (uint32_t)ANI_MIN_ANCHORS) {
                    if (rl == wl) tried = 1u;
                } else {
                    cur = wl;
                    for (;;) {
                        if (rr_from(used, rbase + cur)) break;
                        if (rl == cur) used = 1u;
                        const uint32_t p = rr_from(pred, rbase + cur);
                        if (!p) break;
                        cur = p - 1u;
                    }
                    ChainRec cr;
                    cr.score = (int32_t)rr_from((uint32_t)f, rbase + wl);
                    cr.n = n;
                    cr.n_seeds = rr_from(qi_last, rbase + wl) - rr_from(qi_first, rbase + first) + 1u;
                    cr.q0 = rr_from(q_first, rbase + first); cr.q1 = rr_from(q_last, rbase + wl);
                    {   // a predecessor lies strictly behind on the other genome too: the chain's extent there is spanned by its two ends
                        const uint32_t re = rr_from(rr_last, rbase + wl) & HIT_POS_MASK, rb = rr_from(r_first, rbase + first);
                        cr.r0 = re < rb ? re : rb; cr.r1 = re > rb ? re : rb;
                    }
                    cr.chunk = c;
                    if (rl == 0u) {
                        if (nfin < FAST_SLOTS) slots[nfin] = cr;
                        else {                                   // beyond the chunk's slots: the pair's list (what the general kernel fills)
                            const uint32_t slot = atomicAdd(&pair_nch[pi], 1u);
                            if (slot < pd.c_cap) chains[pd.c_base + slot] = cr;
                            else atomicOr(&flags[0], 8u);
                        }
                    }
                    nfin++;
                }
            }
        }
        // ---- results: the chunk's state, its anchors; declined chunks on to the general kernel (one atomic per wavefront)
        const bool lead = live && rl == 0u;
        if (lead && !cplx) {
            chunk_state[t] = nfin < FAST_SLOTS ? nfin : FAST_SLOTS;
            if (ia) atomicAdd(&pair_na[pi], ia);
        }
        const unsigned long long dm = __ballot(lead && cplx);
        if (dm) {
            const uint32_t leader = (uint32_t)__ffsll((long long)dm) - 1u;
            uint32_t base = 0;
            if (lane == leader) base = atomicAdd(next_count, (uint32_t)__popcll(dm));
            base = (uint32_t)__shfl((int)base, (int)leader, 64);
            if (lead && cplx) { chunk_state[t] = CHUNK_SLOW; next_list[base + (uint32_t)__popcll(dm & ((1ull << lane) - 1ull))] = t; }
            if (stats) {
                for (uint32_t cz = 2; cz <= 9u; cz++) {
                    const unsigned long long cm = __ballot(lead && cplx && cause == cz);
                    if (cm && lane == 0) atomicAdd(stats + cz, (uint32_t)__popcll(cm));
                }
            }
        }
        if (stats) {
            const unsigned long long sm = __ballot(lead && !cplx);
            if (sm && lane == 0) atomicAdd(stats + 0, (uint32_t)__popcll(sm));
        }
    }
}

void launch_chain_rruns(hipStream_t st, unsigned grid, SetView A, SetView B, const PairDesc *pairs, const uint32_t *list, const uint32_t *n_ptr,
                        const uint32_t *gen_list, const uint32_t *gen_cnt, uint32_t gen_cap, const RunRec *recs, const uint32_t *chunk_rec0,
                        const uint4 *multi, ChainRec *fast_chains, uint32_t *chunk_state, ChainRec *chains, uint32_t *pair_nch, uint32_t *pair_na,
                        uint32_t *next_list, uint32_t *next_count, uint32_t *stats, uint32_t *flags, const uint32_t *chunk_pair)
{
    hipLaunchKernelGGL(chain_rruns_kernel, dim3(grid), dim3(64 * RR_WAVES), 0, st, A, B, pairs, list, n_ptr, gen_list, gen_cnt, gen_cap, recs,
                       chunk_rec0, multi, fast_chains, chunk_state, chains, pair_nch, pair_na, next_list, next_count, stats, flags, chunk_pair);
}
