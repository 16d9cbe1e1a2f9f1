// chain_extract.hip -- run records from the hit words (run_extract_kernel) and the SIEVE over them (chain_single_kernel)
#include "chain.h"

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
#ifndef SIEVE_SORT
#define SIEVE_SORT 0            // 1: a workgroup's chunks dealt to its lanes by record count (below).  Measured in round 5 and left off: the
#endif                          // wavefronts finish together, but neighbouring chunks' records no longer load together -- 448 against 425 us per launch
                                // on the real-structure set, no change on the benchmark (profiles/run/r5_sieve.sh)
#define SIEVE_CLASSES 10u
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
                                                           uint32_t *__restrict__ counters, uint4 *__restrict__ gen_list,
                                                           uint32_t *__restrict__ gen_cnt, uint32_t gen_cap,
                                                           uint32_t *__restrict__ pair_na, int xcd_remap, uint32_t *__restrict__ chunk_pair)
{
    // workgroups in launch order (dealt round-robin to the 8 XCDs): every record is read once, there is nothing an XCD's L2
    // could share, and one contiguous stream over the chip measured 1.8 ms per step faster than an eighth of the list per XCD
    const uint32_t wg = blockIdx.x;
#if SIEVE_SORT
    // The loop below takes as many trips as a chunk has records (1, 3, 5 ... with 0, 1, 2 strays), and a wavefront as many as its
    // longest chunk: 21 of 64 lanes at work on the benchmark.  A chunk's record count is (nearly always) the distance to the next
    // chunk's first record, known before any record is read: the workgroup's 256 chunks are dealt to its lanes in the order of that
    // count (a counting sort over ten classes through LDS), so that the chunks of a wavefront finish together.  Which lane takes
    // which chunk changes nothing a chunk computes.
    __shared__ uint32_t s_cnt[4][SIEVE_CLASSES], s_off[4][SIEVE_CLASSES], s_t[256], s_i[256];
    uint32_t t, idx0_pre;
    {
        const uint32_t t0 = wg * 256u + threadIdx.x, lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
        uint32_t key = SIEVE_CLASSES - 1u, i0 = 0xFFFFFFFFu;            // beyond the last chunk: behind everything
        if (t0 < total_chunks) {
            i0 = chunk_rec0[t0];
            const uint32_t i1 = t0 + 1u < total_chunks ? chunk_rec0[t0 + 1u] : 0xFFFFFFFFu;
            const uint32_t d = i1 - i0;
            key = i0 == 0xFFFFFFFFu ? 0u : (i1 != 0xFFFFFFFFu && d - 1u < SIEVE_CLASSES - 3u) ? d : SIEVE_CLASSES - 2u;   // unknown (the pair's or quarter's last chunk): with the long ones
        }
        uint32_t rank = 0;
#pragma unroll
        for (uint32_t v = 0; v < SIEVE_CLASSES; v++) {
            const unsigned long long m = __ballot(key == v);
            if (lane == 0) s_cnt[wv][v] = (uint32_t)__popcll(m);
            if (key == v) rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        }
        __syncthreads();
        if (threadIdx.x < 4u * SIEVE_CLASSES) {         // exclusive prefix over (class, wavefront) pairs, class-major
            const uint32_t v = threadIdx.x / 4u, w = threadIdx.x & 3u;
            uint32_t o = 0;
            for (uint32_t vv = 0; vv < SIEVE_CLASSES; vv++)
                for (uint32_t ww = 0; ww < 4u; ww++)
                    o += (vv < v || (vv == v && ww < w)) ? s_cnt[ww][vv] : 0u;
            s_off[w][v] = o;
        }
        __syncthreads();
        const uint32_t at = s_off[wv][key] + rank;
        s_t[at] = t0; s_i[at] = i0;
        __syncthreads();
        t = s_t[threadIdx.x]; idx0_pre = s_i[threadIdx.x];
    }
#else
    const uint32_t t = wg * 256u + threadIdx.x;
#endif
    const bool in = t < total_chunks;
    uint32_t pi = 0, n_add = 0, slow_why = 0;
    bool to_gen = false, to_slow = false;
    uint4 gd0 = make_uint4(0, 0, 0, 0), gd1 = gd0;       // what the run loop needs of a chunk it is handed (chain.h, GenItem)
    if (in) {
#if SIEVE_SORT
        const uint32_t idx0 = idx0_pre;
#else
        const uint32_t idx0 = chunk_rec0[t];                 // independent of the descriptor: in flight beside it
#endif
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
            to_slow = true; slow_why = ((pd.flags & 8u) || (xcd_remap & 2)) ? 6u : 8u;
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
            uint32_t p_hw[SIEVE_PATHS] = {0, 0, 0}, p_G[SIEVE_PATHS] = {0, 0, 0};     // closed and current paths: key, first diagonal, steps
            int32_t p_D[SIEVE_PATHS] = {0, 0, 0};
            uint32_t p_lq[SIEVE_PATHS] = {0, 0, 0};                                   // ... and the position of their last anchor
            ChainRec *slots = fast_chains + t;          // slot k of chunk t lies at [k * total_chunks + t] (chain.h)
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
                    slots[(uint64_t)(nfin++) * total_chunks] = cr;                                                                               \
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
                                if (npath + 1u >= SIEVE_PATHS) { fail = true; SIEVE_WHY(10); break; }
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
            else {
                to_gen = true; chunk_pair[t] = pi;
                gd0 = make_uint4(t, pi, idx0, s1); gd1 = make_uint4(pd.rec_base, pd.multi_base, c, s0);
            }
        }
    }
    {
        // straight to the general kernel: one atomic per wavefront (a single counter takes one every ~10 ns)
        const unsigned long long sm = __ballot(to_slow);
        if (sm) {
            const uint32_t ln = threadIdx.x & 63u, leader = (uint32_t)__ffsll((long long)sm) - 1u;
            uint32_t base = 0;
            if (ln == leader) base = atomicAdd(counters, (uint32_t)__popcll(sm));
            base = (uint32_t)__shfl((int)base, (int)leader, 64);
            if (to_slow) slow_list[base + (uint32_t)__popcll(sm & ((1ull << ln) - 1ull))] = t;
            const unsigned long long s6 = __ballot(to_slow && slow_why == 6u);
            if (ln == 0) { if (s6) atomicAdd(counters + 7, (uint32_t)__popcll(s6)); if (sm & ~s6) atomicAdd(counters + 9, (uint32_t)__popcll(sm & ~s6)); }
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
            if (to_gen) {
                uint4 *gi = gen_list + 2u * ((uint64_t)li * gen_cap + base + (uint32_t)__popcll(gm & ((1ull << lane) - 1ull)));
                gi[0] = gd0; gi[1] = gd1;
            }
        }
    }
    // anchors of the pairs: one atomic per wavefront and pair (a wavefront's chunks belong to one pair, or since the lanes are dealt out by
    // record count to the two or three pairs of the workgroup's window)
    {
        unsigned long long todo = __ballot(in && n_add != 0u);
        while (todo) {
            const uint32_t first = (uint32_t)__ffsll((long long)todo) - 1u;
            const uint32_t p0 = (uint32_t)__shfl((int)pi, (int)first, 64);
            const bool mine = in && n_add != 0u && pi == p0;
            uint32_t v = mine ? n_add : 0u;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
            if ((threadIdx.x & 63u) == 0 && v) atomicAdd(&pair_na[p0], v);
            todo &= ~__ballot(mine);
        }
    }
}

void launch_run_extract(hipStream_t st, unsigned grid, SetView A, SetView B, const PairDesc *pairs, const uint32_t *hits, RunRec *recs,
                        uint32_t *pair_over, uint32_t *chunk_rec0)
{
    hipLaunchKernelGGL(run_extract_kernel, dim3(grid), dim3(256), 0, st, A, B, pairs, hits, recs, pair_over, chunk_rec0);
}
void launch_chain_single(hipStream_t st, unsigned grid, SetView A, SetView B, const PairDesc *pairs, uint32_t npairs, uint32_t total_chunks,
                         const RunRec *recs, const uint32_t *pair_over, const uint32_t *chunk_rec0, const uint32_t *wg_pair, const uint4 *multi,
                         ChainRec *fast_chains, uint32_t *chunk_state, uint32_t *slow_list, uint32_t *counters, uint4 *gen_list,
                         uint32_t *gen_cnt, uint32_t gen_cap, uint32_t *pair_na, int xcd_remap, uint32_t *chunk_pair)
{
    hipLaunchKernelGGL(chain_single_kernel, dim3(grid), dim3(256), 0, st, A, B, pairs, npairs, total_chunks, recs, pair_over, chunk_rec0, wg_pair, multi,
                       fast_chains, chunk_state, slow_list, counters, gen_list, gen_cnt, gen_cap, pair_na, xcd_remap, chunk_pair);
}
