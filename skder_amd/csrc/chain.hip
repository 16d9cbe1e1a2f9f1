// chain.hip -- host orchestration of the chaining stage: anchors, chunked chaining, ANI / aligned fraction for a list of genome pairs.
//
// Device restatement of oracle/ani_oracle.c oracle_pair() (steps 1-6); integer results are bit-identical by construction, the
// few double operations are + - * / in the oracle's order (every file is compiled with -ffp-contract=off).  The kernels, by stage:
//   chain_join.hip      join_probe_kernel    hit word of every (pair, seed of the chunked genome), the probed genome's index in LDS
//   chain_extract.hip   run_extract_kernel   seed-parallel: runs of seeds that continue the previous hit -> run records
//                       chain_single_kernel  the SIEVE: one lane per (pair, 20 kb chunk), a chunk of plain paths and two strays IS its chains
//   chain_runs.hip      chain_runs_kernel    the RUN LOOP: one lane per remaining chunk, banded DP over run records against a 4-run ring,
//                                            exact for the chunks it can prove
//   chain_rows.hip      chain_rows_kernel    the GENERAL kernel: one 16-lane row per remaining chunk, the unabridged algorithm in LDS
//   chain_slow.hip      slow_wave_kernel, slow_caps / slow_anchors / slow_chain: what does not fit a row (one wavefront per chunk; global memory)
//   chain_finalize.hip  finalize_kernel_t    one workgroup per pair: overlap filter, sums, the two roots, the model, aligned fractions
#include <algorithm>
#include <chrono>
#include <thread>
#include <type_traits>
#include <atomic>

// which ANI the edge records (and the tables written from them) carry: 0 = after the learned-ANI stand-in (default), 1 = the raw k-mer
// estimate -- skani's `--no-learned-ani` (skder_amd_set_ani_output, api.hip)
std::atomic<int> g_ani_output_raw{0};

#include "chain.h"


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
    DevBuf<uint32_t> pair_over, chunk_rec0, wg_pair, gen_cnt, chunk_pair, rows_next;
    DevBuf<uint4> gen_list;
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
    uint32_t fin_need = 0;                       // chains per pair the finalize step was last retried for (chain_stage sizes its LDS by it)
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
    hipStream_t queues[3] = {st, st, st};
    if (const char *e = getenv("SKDER_AMD_CHUNK_BUDGET")) budget = strtoull(e, nullptr, 10);
    ChainWork &W = *chain_work(ctx);
    // which ANI the records of THIS call carry (skder_amd_set_ani_output): read once, so that a call's records are of one kind even if
    // another thread flips the switch while it runs
    const bool ani_raw_out = g_ani_output_raw.load(std::memory_order_relaxed) != 0;
    const SetView VA = view_of(SA), VB = view_of(SB);
    // debugging switches: SKDER_AMD_FORCE_SLOW sends every chunk down the slow path;
    // SKDER_AMD_NO_SIEVE hands every chunk with hits to chain_runs_kernel (to tell the two fast kernels apart behind a parity failure)
    int xcd_remap = 1 | (getenv("SKDER_AMD_FORCE_SLOW") ? 2 : 0);
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
        HIPCHECK(hipMemsetAsync(S.counters.p, 0, 256, S.st));
        HIPCHECK(hipMemsetAsync(S.flags.p, 0, 64, S.st));
        HIPCHECK(hipEventRecord(S.ev[1], S.st));
        if (S.nchunks) {
            HIPCHECK(hipMemsetAsync(S.chunk_rec0.p, 0xFF, S.nchunks * 4, S.st));
            HIPCHECK(hipMemsetAsync(S.pair_over.p, 0, nb * 4, S.st));
            launch_run_extract(S.st, nb, VA, VB, S.d_pairs.p, S.hits.p, S.recs.p, S.pair_over.p, S.chunk_rec0.p);
            HIPCHECK(hipEventRecord(S.ev[6], S.st));
            const unsigned nwg = (unsigned)((S.nchunks + 255) / 256);
            const uint32_t gen_cap = ((nwg + GEN_LISTS - 1u) / GEN_LISTS) * 256u;
            HIPCHECK(hipMemsetAsync(S.gen_cnt.p, 0, GEN_LISTS * 4, S.st));
            launch_chain_single(S.st, nwg, VA, VB, S.d_pairs.p, nb, (uint32_t)S.nchunks, S.recs.p,
                               S.pair_over.p, S.chunk_rec0.p, S.wg_pair.p, S.multi.p, S.fast_chains.p, S.chunk_state.p, S.slow_list.p, S.counters.p,
                               S.gen_list.p, S.gen_cnt.p, gen_cap, S.pair_na.p, xcd_remap, S.chunk_pair.p);
            // what the sieve left: the run loop (one lane per chunk, a ring of four runs); what it gives up goes to the general kernel.
            launch_chain_runs(S.st, nwg < 1024u ? nwg : 1024u, S.gen_list.p, S.gen_cnt.p, gen_cap, S.recs.p, S.multi.p, S.fast_chains.p, (uint32_t)S.nchunks, S.chunk_state.p,
                              S.counters.p, S.pair_na.p, S.slow_list.p, S.counters.p, S.counters.p + 40);
        } else {
            HIPCHECK(hipEventRecord(S.ev[6], S.st));
        }
        HIPCHECK(hipEventRecord(S.ev[2], S.st));
        // declined chunks: one wavefront each, in LDS (count read on the device); the rare chunk with more
        // than 1024 anchors is put on over_list and dealt with after the batch's results are back
        if (S.nchunks) {
            // the chunks the fast path left: one 16-lane row each (chain_rows.hip); what does not fit a row goes on to one
            // wavefront each, and from there (more than SLOWW_MAXA anchors) to the global-memory kernels behind the batch
            static const bool no_rows = getenv("SKDER_AMD_NO_ROWS") != nullptr;        // (A/B: the wavefront kernel for everything)
            const uint32_t *wave_list = S.slow_list.p, *wave_count = S.counters.p;
            if (!no_rows) {
                const uint64_t wantr = (S.nchunks + ROWS_WAVES * 4 - 1) / (ROWS_WAVES * 4);
                launch_chain_rows(S.st, (unsigned)(wantr < 8192 ? wantr : 8192), VA, VB, S.d_pairs.p, S.slow_list.p, S.counters.p, S.hits.p, S.multi.p,
                                  S.chains.p, S.pair_nch.p, S.pair_na.p, S.rows_next.p, S.counters.p + 32, S.flags.p, S.chunk_pair.p);
                wave_list = S.rows_next.p; wave_count = S.counters.p + 32;
            }
            const uint64_t want = (S.nchunks + SLOWW_WAVES - 1) / SLOWW_WAVES;
            launch_slow_wave(S.st, (unsigned)(want < 2048 ? want : 2048), VA, VB, S.d_pairs.p, nb, wave_list, wave_count, S.hits.p, S.multi.p, S.chains.p, S.pair_nch.p, S.pair_na.p, S.over_list.p,
                               S.counters.p + 15, S.flags.p, S.chunk_pair.p);
        }
        HIPCHECK(hipEventRecord(S.ev[3], S.st));
        // LDS capacity of the finalize step: the most chains any pair of the batch can plausibly have
        // (1.5 per chunk + slack), rounded up; a batch in which some pair has more is finalized again with
        // the full 4096 (140 KB) when its results are read
        uint32_t max_chunks = 0;
        for (const PairDesc &d : S.hp) max_chunks = d.n_chunks > max_chunks ? d.n_chunks : max_chunks;
#ifdef FIN_CAP_POW2
        uint32_t lds_cap = 512;
        while (lds_cap < 3u * max_chunks / 2 + 128 && lds_cap < 4096) lds_cap <<= 1;   // retried with more if a pair needs it
#else
        // (not rounded up to a power of two: the kernel's occupancy is bound by this LDS, 35 bytes per chain)
        uint32_t lds_cap = (3u * max_chunks / 2 + 128 + 63u) & ~63u;                   // retried with more if a pair needs it
        if (lds_cap < 256u) lds_cap = 256u;
        // genomes whose pairs hold more chains than that (fragmented assemblies): what the last batch that ran over needed, up to twice the estimate
        if (W.fin_need > lds_cap) lds_cap = W.fin_need < 2u * lds_cap ? W.fin_need : 2u * lds_cap;
        if (lds_cap > 4096u) lds_cap = 4096u;
#endif
        S.lds_cap = lds_cap;
        if (nb)
            launch_finalize(S.st, nb, lds_cap, VA, VB, S.d_pairs.p, S.fast_chains.p, (uint32_t)S.nchunks, S.chunk_state.p, S.chains.p, S.pair_nch.p, S.pair_na.p, S.d_out.p,
                            S.flags.p, S.chunk_mark.p);
        HIPCHECK(hipGetLastError());     // a rejected launch (resources) must not pass as an empty result
        HIPCHECK(hipEventRecord(S.ev[4], S.st));
        HIPCHECK(hipMemcpyAsync(S.h_out, S.d_out.p, nb * sizeof(PairOut), hipMemcpyDeviceToHost, S.st));
        HIPCHECK(hipMemcpyAsync(S.h_cnt, S.counters.p, 64, hipMemcpyDeviceToHost, S.st));
        HIPCHECK(hipMemcpyAsync(S.h_cnt + 16, S.flags.p, 4, hipMemcpyDeviceToHost, S.st));
        HIPCHECK(hipMemcpyAsync(S.h_cnt + 17, S.counters.p + 32, 4, hipMemcpyDeviceToHost, S.st));      // [17] chunks on to the wavefront kernel
#ifndef SKDER_SIEVE_STATS
        HIPCHECK(hipMemcpyAsync(S.h_cnt + 24, S.flags.p + 8, 32, hipMemcpyDeviceToHost, S.st));      // (filled by builds with SKDER_ROWS_STATS / SKDER_SLOW_STATS)
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
        grow(S.counters, 64); grow(S.flags, 16); grow(S.rows_next, nchunks + 1);
        grow(S.pair_na, nb); grow(S.pair_nch, nb); grow(S.pair_nmulti, nb);
        grow(S.hits, nhits + 64); grow(S.multi, nmulti + 1);
        grow(S.recs, nrecs + 8); grow(S.pair_over, nb + 1); grow(S.chunk_rec0, nchunks + 1); grow(S.gen_list, 2 * (nchunks + 256ull * GEN_LISTS + 1)); grow(S.gen_cnt, GEN_LISTS);
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
                join_probe_allow_large_lds();
                finalize_allow_large_lds();
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
            uint32_t join_smem = (uint32_t)(want < JOIN_SMEM_MAX ? want : JOIN_SMEM_MAX) / 64u * 64u + 64u;
            if (want <= JOIN_SMEM_TWO && join_smem > JOIN_SMEM_TWO) join_smem = JOIN_SMEM_TWO;
            if (getenv("SKDER_AMD_DEBUG")) {
                const int per_cu = join_probe_resident_per_cu(join_smem);
                fprintf(stderr, "[skder_amd] join: %zu workgroups of %u threads, %u bytes of LDS each: %d resident per CU (runtime's answer)\n", hg.size(), JOIN_THREADS, join_smem, per_cu);
            }
            launch_join_probe(S.st_join, (unsigned)hg.size(), join_smem, VA, VB, S.d_pairs.p, reinterpret_cast<const JoinGroup *>(S.groups.p), S.hits.p,
                              S.multi.p, S.pair_nmulti.p);
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
#ifndef SKDER_SIEVE_STATS
        if (S.h_cnt[29] && getenv("SKDER_AMD_DEBUG"))
        fprintf(stderr, "[skder_amd] general path: %u chunks, %u anchors, %u full look-backs (%u passes), %u stretches, %u chain ends, %u ladder visits; %u chunks on to the wavefront kernel\n", S.h_cnt[29], S.h_cnt[24], S.h_cnt[25], S.h_cnt[30], S.h_cnt[26], S.h_cnt[27], S.h_cnt[28], S.h_cnt[17]);
#endif
#ifdef FIN_TIMING
        { void finalize_timing_dump(); if (getenv("SKDER_AMD_DEBUG")) finalize_timing_dump(); }
#endif
        if (getenv("SKDER_AMD_DEBUG")) {
            const uint32_t *hcnt = S.h_cnt;
            fprintf(stderr, "[skder_amd] batch: over %u; %u pairs %llu chunks, room for %llu run records, %u to the run loop, %u on to the general kernel (none %u, slots %u, hits %u, ring %u, branch %u, score %u, qrep %u, inside %u, records-full %u, run-not-dominant %u)\n",
                    nover, nb, (unsigned long long)S.nchunks, (unsigned long long)S.nrecs, hcnt[11], nslow, hcnt[1], hcnt[2], hcnt[3], hcnt[4], hcnt[5], hcnt[6], hcnt[7], hcnt[8], hcnt[9], hcnt[10]);
        }
        // Rare-path fix-ups; the slot's buffers are untouched since (the batches in flight use the other slots).
        //  * nover: chunks the wave kernel could not hold go through the global-memory kernels;
        //  * flag 8: a pair produced more slow-path chains than its region holds (repeats: one chunk can
        //    chain to every copy).  pair_nch keeps counting past the capacity, so after a complete attempt it
        //    holds the number wanted: every pair gets exactly that and the chaining stage runs again;
        //  * flag 16: a pair has more chains than the LDS capacity chosen for finalize: again at 4096.
        auto finalize_and_fetch = [&]() {
            launch_finalize(S.st, nb, S.lds_cap, VA, VB, S.d_pairs.p, S.fast_chains.p, (uint32_t)S.nchunks, S.chunk_state.p, S.chains.p, S.pair_nch.p, S.pair_na.p, S.d_out.p,
                            S.flags.p, S.chunk_mark.p);
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
                launch_slow_caps(S.st, VA, VB, S.d_pairs.p, nb, S.over_list.p, nover_now, W.cap.p);
                exclusive_scan_u32(W.cap.p, W.abase.p, nover_now + 1, W.ws, S.st);
                uint32_t atotal = 0;
                HIPCHECK(hipMemcpyAsync(&atotal, W.abase.p + nover_now, 4, hipMemcpyDeviceToHost, S.st));
                HIPCHECK(hipStreamSynchronize(S.st));
                W.a_qi.resize(atotal + 1, S.st); W.a_r.resize(atotal + 1, S.st); W.a_rctg.resize(atotal + 1, S.st);
                W.F.resize(atotal + 1, S.st); W.BP.resize(atotal + 1, S.st); W.ORD.resize(atotal + 1, S.st);
                launch_slow_anchors(S.st, VA, VB, S.d_pairs.p, nb, S.over_list.p, nover_now, W.abase.p, W.a_qi.p, W.a_r.p, W.a_rctg.p, W.slow_n.p, S.flags.p);
                launch_slow_chain(S.st, VA, VB, S.d_pairs.p, nb, S.over_list.p, nover_now, W.abase.p, W.slow_n.p, W.a_qi.p, W.a_r.p, W.a_rctg.p, W.F.p, W.BP.p,
                                  W.ORD.p, S.chains.p, S.pair_nch.p, S.pair_na.p, S.flags.p);
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
                // room for the pair with the most chains (the kernel reported what each such pair wants), not at once the full 4096 (140 KB: one pair per CU)
                uint32_t want = 0;
                for (uint32_t i = 0; i < nb; i++)
                    if (S.h_out[i].n_chains == 0xFFFFFFFFu && S.h_out[i].n_chains_all > want) want = S.h_out[i].n_chains_all;
                uint32_t cap = want > 4032u ? 4096u : (want + 63u) & ~63u;
                if (cap <= S.lds_cap) cap = S.lds_cap + 64u;
                W.fin_need = cap;
                S.lds_cap = cap;
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
                    launch_finalize_global(S.st, (unsigned)ng, VA, VB, S.d_pairs.p, S.fast_chains.p, (uint32_t)S.nchunks, S.chunk_state.p, S.chains.p, S.pair_nch.p, S.pair_na.p,
                                           S.d_out.p, S.flags.p, S.chunk_mark.p, gws.p, d_off.p, d_list.p, d_cap.p);
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
            e.ani = ani_raw_out ? o.ani_raw : o.ani;
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

void rectangle_impl(skder_sketches *refs, skder_sketches *queries, double screen_pct, const uint8_t *live_refs)
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
        if (live_refs) {      // the caller has no use for rows of the other reference genomes: their candidate pairs are not chained
            size_t w = 0;
            for (size_t k = 0; k < ppart.size(); k++)
                if (live_refs[ppart[k]]) { ppart[w] = ppart[k]; prow[w] = prow[k]; w++; }
            ppart.resize(w); prow.resize(w);
        }
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
