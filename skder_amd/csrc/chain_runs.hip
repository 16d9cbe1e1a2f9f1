// chain_runs.hip -- the RUN LOOP of the chaining stage: the chunks the sieve (chain_single_kernel, chain_extract.hip) could not settle,
// one lane each, a banded DP over the chunk's run records against a ring of four runs; what it cannot prove goes on to the
// general kernel (chain_rows.hip).  Since round 5 the kernel is a loop of SERVICE + ROUND per wavefront: a lane whose chunk is finished
// waits; when enough lanes are free the wavefront writes their chains out, appends the declined chunks to its LDS buffer and hands
// every free lane the next chunk of the wavefront's range of the work list (the sieve's 32-byte items), so that the rounds run with
// most lanes at work instead of as many as the longest of 64 chunks leaves.
#include "chain.h"

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
// goes to the slow path.  Rounds are uniform across the wavefront: every lane that holds a chunk takes one anchor per round.
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

__global__ __launch_bounds__(256) void chain_runs_kernel(const uint4 *__restrict__ gen_list, const uint32_t *__restrict__ gen_cnt,
                                                         uint32_t gen_cap, const RunRec *__restrict__ recs, const uint4 *__restrict__ multi,
                                                         ChainRec *__restrict__ fast_chains, uint32_t fast_stride, uint32_t *__restrict__ chunk_state,
                                                         uint32_t *__restrict__ slow_count, uint32_t *__restrict__ pair_na,
                                                         uint32_t *__restrict__ decl_list, uint32_t *__restrict__ decl_count,
                                                         uint32_t *__restrict__ work_next, uint32_t refill_min)
{
    // the chunks chain_single_kernel could not settle, one per lane; their number is only known on the device: the GEN_LISTS
    // lists are laid end to end (offsets by a scan of the 256 counts, in LDS) and the wavefronts of a fixed grid DRAW their chunks
    // from one counter.  A lane keeps its chunk for as many rounds as the chunk has anchors to place, and chunks differ (3 to 60
    // records; a declined chunk stops at once): with 64 chunks handed out together a wavefront ran until its longest chunk was
    // through, 11 - 23 of its lanes at work per round (round 4's counters: 12 of 64 lanes per VALU instruction).  Now lanes whose
    // chunk is finished wait until `refill_min` of them are free; then the wavefront writes their results and hands them new chunks
    // in one pass (the service block below), so that the rounds run with at least 64 - refill_min lanes at work until the lists run dry.
    __shared__ uint32_t g_off[GEN_LISTS + 1], g_ws[4];
    __shared__ uint32_t dbuf_all[4][RUNS_DECL_FLUSH + 64];
    {
        uint32_t total;
        const uint32_t ex = block_excl_scan_256(gen_cnt[threadIdx.x], g_ws, total);
        g_off[threadIdx.x] = ex;
        if (threadIdx.x == 0) g_off[GEN_LISTS] = total;
        __syncthreads();
    }
    const uint32_t n_items = g_off[GEN_LISTS];
    if (blockIdx.x == 0 && threadIdx.x == 0) slow_count[11] = n_items;      // for the host's statistics
    const uint32_t ln = threadIdx.x & 63u;
    // items a wavefront draws at a time: about a quarter of its share of the list, between 64 (a short list still reaches every
    // wavefront: the benchmark's batches leave 50,000 chunks for 4,096 wavefronts) and RUNS_DRAW
    const uint32_t share = n_items / (gridDim.x * 16u);
    const uint32_t draw = share < 64u ? 64u : share > RUNS_DRAW ? RUNS_DRAW : share;
    const int32_t NEG = -0x40000000;
    // ---- a lane's chunk (valid while busy)
    bool busy = false, done = true, cplx = false, exhausted = false;
    uint32_t w_cur = 0, w_end = 0;             // the wavefront's range of the item list (wave-uniform)
    uint32_t cause_acc = 0;                    // lane cz: chunks this wavefront declined for cause cz
    uint32_t *dbuf = dbuf_all[threadIdx.x >> 6];
    uint32_t dn = 0;                           // declined chunks waiting in dbuf (wave-uniform)
    uint32_t t = 0, pi = 0, c = 0, s1 = 0, multi_base = 0, cause = 0u;
    const uint4 *prec = nullptr;
    ChainRec *slots = nullptr;
    Run r0, r1, r2, r3;
    uint32_t ia = 0, nfin = 0, nevict = 0;
    int32_t runmax = NEG;
    // summaries of runs that left the ring: the most recent segment, plus one conservative scalar
    uint32_t s0_seg = 0xFFFFFFFFu, s0_key = 0, s0_q = 0, lost_q = 0;
    int32_t s0_f = NEG, lost_f = NEG, s0_dlo = 0, s0_dhi = 0, lost_dlo = 0, lost_dhi = 0;
    // the keyless summary keeps TWO diagonal intervals (empty: lo > hi): the remnants of the main path and a
    // stray single hit far off its diagonal would otherwise merge into one interval that covers everything in between
    int32_t lost2_dlo = 1, lost2_dhi = 0;
    // record cursor: the chunk's records follow one another in the pair's region, from chunk_rec0 on, in seed order; the
    // record BEHIND a run closes it (a link or the terminator at the end of a quarter), so two records are held and
    // the third is on its way while the first is worked on
    uint32_t idx = 0;
    uint4 a0 = make_uint4(0, 0, 0, 0), a1 = a0, b0 = a0, b1 = a0;
    // One ANCHOR per round and lane: a record's first anchor -- or, for a seed with 2..4 occurrences on the other genome, one of
    // its occurrences per round (pend = occurrences still to come; the lane keeps its record until they are through).  A loop
    // over the occurrences inside the round made the whole wavefront repeat the look-back as often as its most repetitive seed
    // asked: on real genome structure 29 % of the records are such seeds and 82 % of the rounds had one in some lane.
    struct { uint32_t qi, q0, hw, q1, qi1, hw1, n, gsum; } rc;
    rc.qi = 0; rc.q0 = 0; rc.hw = HIT_NONE; rc.hw1 = HIT_NONE; rc.q1 = 0; rc.qi1 = 0; rc.n = 0; rc.gsum = 0;
    uint32_t pend = 0, g0 = HIT_NONE, g1 = HIT_NONE, g2 = HIT_NONE, g3 = HIT_NONE;
    r0.cnt = r1.cnt = r2.cnt = r3.cnt = 0;
    r0.f = r1.f = r2.f = r3.f = NEG;
    r0.q_last = r1.q_last = r2.q_last = r3.q_last = 0; r0.rr_last = r1.rr_last = r2.rr_last = r3.rr_last = 0;
    r0.first_qi = r1.first_qi = r2.first_qi = r3.first_qi = 0; r0.q_first = r1.q_first = r2.q_first = r3.q_first = 0;
    r0.r_pfirst = r1.r_pfirst = r2.r_pfirst = r3.r_pfirst = 0;
    r0.qi_last = r1.qi_last = r2.qi_last = r3.qi_last = 0; r0.idx_last = r1.idx_last = r2.idx_last = r3.idx_last = 0;
    r0.pmax = r1.pmax = r2.pmax = r3.pmax = NEG; r0.r_first = r1.r_first = r2.r_first = r3.r_first = 0;
    r0.seg = r1.seg = r2.seg = r3.seg = 0;
    r0.gs = r1.gs = r2.gs = r3.gs = 0;

#ifdef SKDER_PEAK_STATS      // (measurement build: "best end is not last" split into: the peak's run not in the ring -> 6, a tail of 3 or more -> 8, else 5)
#define PEAK_CAUSE(N, T) ((N) == 0u ? 6u : ((N) == 1u && (T) > 2u) ? 8u : 5u)
#else
#define PEAK_CAUSE(N, T) 5u
#endif
    // (for EMIT_PATH: is ring entry X the run that ends at the peak of the path that ends with E?)
#define PEAK_OF(E, X)                                                                        \
    if (((X).cnt & SUCC_BIT) && (X).first_qi == pk_e_qi && (X).r_pfirst == pk_e_r &&          \
        (((X).rr_last ^ pk_e_rr) & HIT_KEY_MASK) == 0u && (X).f == pk_e_pmax && (X).f > (X).pmax) { \
        pk_n++; pk_cnt = (X).cnt & ~SUCC_BIT; pk_qi1 = (X).qi_last; pk_q1 = (X).q_last; pk_rr = (X).rr_last; pk_f = (X).f; \
    }
#define EMIT_PATH(E)                                                                         \
    do {                                                                                     \
        if ((E).cnt && !((E).cnt & SUCC_BIT) && (E).cnt >= ANI_MIN_ANCHORS) {                \
            if (!((E).f > (E).pmax)) {                                                      \
                /* the path's best end is an EARLIER anchor: the oracle takes the chain that ends THERE first, and what lies behind it is   \
                   left to later ends.  Settled here when at most two anchors lie behind it (they cannot form a chain, whatever their    \
                   order) and the run that ends at the peak is still in the ring: same first anchor (seed, place on the other genome,     \
                   record and strand), marked as having a successor, its score the path's maximum and above everything before it; its     \
                   chain is the prefix.  Anything else -- the peak's run gone, a longer tail, a second candidate -- is declined */         \
                uint32_t pk_n = 0, pk_cnt = 0, pk_qi1 = 0, pk_q1 = 0, pk_rr = 0;                 \
                int32_t pk_f = 0;                                                               \
                const uint32_t pk_e_qi = (E).first_qi, pk_e_r = (E).r_pfirst, pk_e_rr = (E).rr_last; \
                const int32_t pk_e_pmax = (E).pmax;                                             \
                PEAK_OF(E, r0) PEAK_OF(E, r1) PEAK_OF(E, r2) PEAK_OF(E, r3)                       \
                const uint32_t tail_ = (E).cnt - pk_cnt;                                         \
                if (pk_n != 1u || pk_cnt < ANI_MIN_ANCHORS || tail_ < 1u || tail_ > 2u) { cplx = true; cause = PEAK_CAUSE(pk_n, tail_); } \
                else if (nfin >= FAST_SLOTS) { cplx = true; cause = 1; }                         \
                else {                                                                           \
                    ChainRec cr;                                                                 \
                    cr.score = pk_f; cr.n = pk_cnt; cr.n_seeds = pk_qi1 - (E).first_qi + 1;         \
                    cr.q0 = (E).q_first; cr.q1 = pk_q1;                                            \
                    { const uint32_t rl_ = pk_rr & HIT_POS_MASK;                                  \
                      cr.r0 = rl_ < (E).r_pfirst ? rl_ : (E).r_pfirst; cr.r1 = rl_ > (E).r_pfirst ? rl_ : (E).r_pfirst; } \
                    cr.chunk = c;                                                                \
                    slots[(uint64_t)(nfin++) * fast_stride] = cr;                                                          \
                }                                                                                \
            }                                                                                    \
            else if (nfin >= FAST_SLOTS) { cplx = true; cause = 1; }                         \
            else {                                                                           \
                ChainRec cr;                                                                 \
                cr.score = (E).f; cr.n = (E).cnt; cr.n_seeds = (E).qi_last - (E).first_qi + 1; \
                cr.q0 = (E).q_first; cr.q1 = (E).q_last;                                       \
                { /* a predecessor lies strictly behind on the other genome too: the path's extent there is spanned by its two ends */ \
                  const uint32_t rl_ = (E).rr_last & HIT_POS_MASK;                            \
                  cr.r0 = rl_ < (E).r_pfirst ? rl_ : (E).r_pfirst; cr.r1 = rl_ > (E).r_pfirst ? rl_ : (E).r_pfirst; } \
                cr.chunk = c; \
                slots[(uint64_t)(nfin++) * fast_stride] = cr;                                                          \
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

    // (a wavefront's LDS writes are seen by its own later reads: DS operations of a wavefront complete in order)
#define FLUSH_DECLINED()                                                                     \
    do {                                                                                     \
        uint32_t base_ = 0;                                                                  \
        if (ln == 0) base_ = atomicAdd(decl_count, dn);                                      \
        base_ = (uint32_t)__builtin_amdgcn_readfirstlane((int)base_);                        \
        for (uint32_t i_ = ln; i_ < dn; i_ += 64u) decl_list[base_ + i_] = dbuf[i_];          \
        dn = 0;                                                                              \
    } while (0)

    for (;;) {
        // ---- service: results of finished chunks out, new chunks in -- when enough lanes are free, or nothing else is left to do
        {
            const bool fin = busy && done && pend == 0u;
            const unsigned long long finm = __ballot(fin), idlem = __ballot(!busy);
            const bool none_active = (finm | idlem) == ~0ull;
            const uint32_t n_free = (uint32_t)__popcll(finm | idlem);
            if (none_active || (finm && (uint32_t)__popcll(finm) >= refill_min) || (!exhausted && n_free >= refill_min)) {
                if (fin && !cplx) EMIT_PATH(r3);
                if (fin && !cplx) EMIT_PATH(r2);
                if (fin && !cplx) EMIT_PATH(r1);
                if (fin && !cplx) EMIT_PATH(r0);
                // declined chunks: ONE atomic per wavefront and service for the list (a counter shared by the whole device takes an
                // atomic every ~10 ns: a million lanes adding one each were the kernel's time), the lanes take consecutive places;
                // the per-cause statistics likewise
                const bool decl = fin && cplx;
                const unsigned long long dm = __ballot(decl);
                if (dm) {
                    // declined chunks collect in the wavefront's LDS buffer and go to the shared list RUNS_DECL_FLUSH or more at a time:
                    // ONE atomic on the list's counter per flush (a counter shared by the whole device takes an atomic every ~10 ns,
                    // serialised: one per chunk -- round 3 -- or per service made the counter the kernel's time)
                    if (decl) { chunk_state[t] = CHUNK_SLOW; dbuf[dn + (uint32_t)__popcll(dm & ((1ull << ln) - 1ull))] = t; }
                    dn += (uint32_t)__popcll(dm);
                    // per-cause statistics: lane cz keeps the wavefront's count of cause cz and adds it to the shared counters ONCE, when the
                    // wavefront is through (the counters share one cache line with decl_count)
                    for (uint32_t cz = 1; cz <= 10u; cz++) {
                        const unsigned long long cm = __ballot(decl && cause == cz);
                        if (ln == cz) cause_acc += (uint32_t)__popcll(cm);
                    }
                    if (dn >= RUNS_DECL_FLUSH) { FLUSH_DECLINED(); }
                }
                if (fin && !cplx) {
                    chunk_state[t] = nfin;
                    if (ia) atomicAdd(&pair_na[pi], ia);
                }
                if (fin) busy = false;
                if (!exhausted) {
                    // the wavefront draws its chunks from a range of its own, RUNS_DRAW items at a time: ONE atomic on the shared counter
                    // per range (a device-wide counter takes an atomic every ~10 ns, serialised: one per service made the counter the
                    // kernel's time -- 15.5 ms at refill_min 16 where 64 took 7.8, measured)
                    const unsigned long long freem = __ballot(!busy);
                    const uint32_t nfree = (uint32_t)__popcll(freem);
                    if (w_cur >= w_end) {
                        uint32_t base = 0;
                        if (ln == 0) base = atomicAdd(work_next, draw);
                        w_cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                        w_end = w_cur + draw < n_items ? w_cur + draw : n_items;
                        if (w_cur >= n_items) { exhausted = true; w_cur = w_end = n_items; }
                    }
                    const uint32_t w = w_cur + (uint32_t)__popcll(freem & ((1ull << ln) - 1ull));
                    const uint32_t w_lim = w_end;
                    w_cur = w_cur + nfree < w_end ? w_cur + nfree : w_end;
                    if (!busy && w < w_lim) {
                        uint32_t lo = 0, hi = GEN_LISTS;              // the list that holds item w: last offset <= w
                        while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (g_off[mid] <= w) lo = mid; else hi = mid; }
                        const uint4 *gi = gen_list + 2u * ((uint64_t)lo * gen_cap + (w - g_off[lo]));
                        const uint4 i0 = gi[0], i1 = gi[1];       // (chain.h: what the sieve knew of the chunk)
                        t = i0.x; pi = i0.y; idx = i0.z; s1 = i0.w;
                        multi_base = i1.y; c = i1.z;
                        const uint32_t s0 = i1.w;
                        prec = reinterpret_cast<const uint4 *>(recs + i1.x);
                        slots = fast_chains + t;
                        busy = true; cplx = false; cause = 0u;
                        done = idx == 0xFFFFFFFFu || s1 <= s0;
                        if (!done) { a0 = prec[2u * idx]; a1 = prec[2u * idx + 1u]; b0 = prec[2u * idx + 2u]; b1 = prec[2u * idx + 3u]; }   // a run record is never the last of its quarter
                        r0.cnt = r1.cnt = r2.cnt = r3.cnt = 0;          // cnt == 0: empty ring position
                        r0.f = r1.f = r2.f = r3.f = NEG;
                        r0.pmax = r1.pmax = r2.pmax = r3.pmax = NEG;
                        r0.gs = r1.gs = r2.gs = r3.gs = 0;
                        ia = nfin = nevict = 0; runmax = NEG;
                        s0_seg = 0xFFFFFFFFu; s0_key = 0; s0_q = 0; lost_q = 0;
                        s0_f = NEG; lost_f = NEG; s0_dlo = s0_dhi = lost_dlo = lost_dhi = 0;
                        lost2_dlo = 1; lost2_dhi = 0;
                        pend = 0u; g0 = g1 = g2 = g3 = HIT_NONE;
                    }
                }
                if (!__any(busy)) break;
            }
        }
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
                    const uint4 mv = multi[multi_base + (hw & 0x00FFFFFFu)];
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
    // (straight-line form: every condition of the oracle's look-back loop is evaluated for every lane and applied by selects -- the
    // nested form executed four copies of a six-level branch nest with a third of the lanes in each arm)
    #define TRY(K, E)                                                                                   \
                {                                                                                       \
                    const bool act_ = !exact && !cplx;                                                  \
                    const int32_t dq = qp - (int32_t)(E).q_last;                                        \
                    const bool stop_ = !(E).cnt || best >= runmax + ANI_ANCHOR_SCORE || ia - (E).idx_last > ANI_BAND || dq > ANI_BP_BAND; \
                    const bool look_ = act_ && !stop_ && ((E).rr_last & HIT_KEY_MASK) == key;           \
                    const int32_t rpj = (int32_t)((E).rr_last & HIT_POS_MASK);                           \
                    const int32_t dr = rev ? rpj - rp : rp - rpj;                                       \
                    const int32_t ed = rev ? rpj + (int32_t)(E).q_last : rpj - (int32_t)(E).q_last; /* run diagonal */ \
                    const int32_t off = dg > ed ? dg - ed : ed - dg;                                    \
                    /* an earlier anchor of a run with steps may be in reach where the last one is not */ \
                    const bool far_ = look_ && off > ANI_MAX_GAP && off - (E).gs <= ANI_MAX_GAP;         \
                    const bool near_ = look_ && off <= ANI_MAX_GAP;                                     \
                    /* an INTERIOR anchor of the run could be a valid predecessor where the last one is not */ \
                    const int32_t rf = (int32_t)(E).r_first;                                            \
                    const bool inside = (rev ? rp < rf : rp > rf) && dr <= 0;                           \
                    const bool in_ = near_ && (dq <= 0 || inside);                                      \
                    const int32_t sc = (E).f + ANI_ANCHOR_SCORE - off;                                  \
                    const bool take_ = near_ && !in_ && dr > 0 && dq <= ANI_MAX_LIN && dr <= ANI_MAX_LIN && sc > best; \
                    best = take_ ? sc : best; bj = take_ ? (K) : bj; pgap = take_ ? off : pgap;         \
                    if (far_ || in_) { cplx = true; cause = 7; }                                        \
                    exact = exact || (act_ && stop_);                                                   \
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
    }
    if (dn) { FLUSH_DECLINED(); }
#undef FLUSH_DECLINED
    if (ln >= 1u && ln <= 10u && cause_acc) atomicAdd(slow_count + 1 + ln, cause_acc);
#undef EMIT_PATH
#undef PEAK_OF
#undef EVICT
}


void launch_chain_runs(hipStream_t st, unsigned grid, const uint4 *gen_list, const uint32_t *gen_cnt, uint32_t gen_cap, const RunRec *recs,
                       const uint4 *multi, ChainRec *fast_chains, uint32_t fast_stride, uint32_t *chunk_state, uint32_t *slow_count, uint32_t *pair_na,
                       uint32_t *decl_list, uint32_t *decl_count, uint32_t *work_next)
{
    static const uint32_t refill_min = getenv("SKDER_AMD_RUNS_REFILL") ? (uint32_t)atoi(getenv("SKDER_AMD_RUNS_REFILL")) : RUNS_REFILL_MIN;
    hipLaunchKernelGGL(chain_runs_kernel, dim3(grid), dim3(256), 0, st, gen_list, gen_cnt, gen_cap, recs, multi, fast_chains, fast_stride, chunk_state, slow_count,
                       pair_na, decl_list, decl_count, work_next, refill_min < 1u ? 1u : refill_min > 64u ? 64u : refill_min);
}
