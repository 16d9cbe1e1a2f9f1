// chain.h -- structures and constants shared by the chaining stage's translation units (chain*.hip)
#pragma once
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
#ifndef FIN_BINS
#define FIN_BINS 256
#endif
#ifndef FAST_SLOTS
#define FAST_SLOTS 6         // chain slots per chunk of the two fast kernels (the run loop declines a chunk with more chains).
                             // fast_chains is slot-major: slot k of chunk t at [k * (chunks of the batch) + t], so that the first slots of
                             // neighbouring chunks -- nearly all that is ever written or read -- share cache lines.
                             // MEMORY: 32 bytes per slot and chunk in each of the three work-buffer slots -- at the default budget of
                             // 6 M chunks per batch 1.15 GB per slot set (3.5 GB in all; it was 1.7 GB at three slots per chunk)
#endif
#define SIEVE_PATHS 3        // paths the sieve follows (its own limit; its chains use the first slots)
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

// run records (run_extract_kernel -> sieve, run loop)
struct __attribute__((aligned(16))) RunRec {
    uint32_t qi, q0, hw, cn;      // first seed: index in the chunked genome, position, hit word (or HIT_MULTI | slot, HIT_MANY); hits of the quarter in front of it
    uint32_t pq, pw, pqi, cg;     // the hit in front of it: position, hit word, seed index; diagonal steps of the quarter in front of it
};
#define SEG_SEEDS 256u
#define RUN_GAP 10
#define REC_LINK 0xFFFFFFFEu      // qi of a link record; its q0 is the index of the next record, its hw the first seed of the next quarter
#define REC_END 0xFFFFFFFFu       // qi of the terminator
static_assert(2 * RUN_GAP <= ANI_ANCHOR_SCORE, "run links must keep at least half of the anchor score");

// ---- the general chaining kernel (chain_rows.hip): one 16-lane row per chunk
#ifndef ROWS_MAXA
#define ROWS_MAXA 160          // anchors a row holds in LDS (measured: 128 / 160 / 192 / 256 -> 5.1 / 4.8 / 5.2 / 7.1 ms on the real-structure set: occupancy against fall-through)
#endif
#ifndef ROWS_WAVES
#define ROWS_WAVES 2           // wavefronts (of four rows) per workgroup
#endif
void launch_chain_rows(hipStream_t st, unsigned grid, SetView A, SetView B, const PairDesc *pairs, const uint32_t *list, const uint32_t *n_ptr,
                       const uint32_t *hits, const uint4 *multi, ChainRec *chains, uint32_t *pair_nch, uint32_t *pair_na, uint32_t *next_list,
                       uint32_t *next_count, uint32_t *flags, const uint32_t *chunk_pair);

// ---- the run loop (chain_runs.hip)
#define GEN_LISTS 256u
// an item of the run loop's lists: two uint4 -- (chunk, pair, first record, end of the chunk's seeds), (the pair's record region,
// its multi-occurrence lists, the chunk's number inside the pair, first seed): the sieve has them in registers when it passes a
// chunk on, and the run loop starts a chunk with ONE dependent load instead of four
void launch_chain_runs(hipStream_t st, unsigned grid, const uint4 *gen_list, const uint32_t *gen_cnt, uint32_t gen_cap, const RunRec *recs,
                       const uint4 *multi, ChainRec *fast_chains, uint32_t fast_stride, uint32_t *chunk_state, uint32_t *slow_count, uint32_t *pair_na,
                       uint32_t *decl_list, uint32_t *decl_count, uint32_t *work_next);
#define RUNS_DRAW 512u          // most items a wavefront takes from the shared counter at a time
#define RUNS_DECL_FLUSH 64u     // declined chunks a wavefront collects before it appends them to the shared list
#define RUNS_REFILL_MIN 24u     // lanes of a wavefront that must be free before finished chunks are written out and new ones handed over

// ---- the join (chain_join.hip)
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

void launch_join_probe(hipStream_t st, unsigned grid, uint32_t smem, SetView A, SetView B, const PairDesc *pairs, const JoinGroup *groups,
                       uint32_t *hits, uint4 *multi, uint32_t *pair_nmulti);
void join_probe_allow_large_lds();
int join_probe_resident_per_cu(uint32_t smem);

// ---- run records and the sieve (chain_extract.hip)
void launch_run_extract(hipStream_t st, unsigned grid, SetView A, SetView B, const PairDesc *pairs, const uint32_t *hits, RunRec *recs,
                        uint32_t *pair_over, uint32_t *chunk_rec0);
void launch_chain_single(hipStream_t st, unsigned grid, SetView A, SetView B, const PairDesc *pairs, uint32_t npairs, uint32_t total_chunks,
                         const RunRec *recs, const uint32_t *pair_over, const uint32_t *chunk_rec0, const uint32_t *wg_pair, const uint4 *multi,
                         ChainRec *fast_chains, uint32_t *chunk_state, uint32_t *slow_list, uint32_t *counters, uint4 *gen_list,
                         uint32_t *gen_cnt, uint32_t gen_cap, uint32_t *pair_na, int xcd_remap, uint32_t *chunk_pair);

// ---- the fall-through tiers (chain_slow.hip)
#ifndef SLOWW_MAXA
#define SLOWW_MAXA 384
#endif
#define SLOWW_WAVES 4        // wavefronts (chunks) per workgroup
void launch_slow_wave(hipStream_t st, unsigned grid, SetView A, SetView B, const PairDesc *pairs, uint32_t npairs, const uint32_t *list,
                      const uint32_t *n_ptr, const uint32_t *hits, const uint4 *multi, ChainRec *chains, uint32_t *pair_nch, uint32_t *pair_na,
                      uint32_t *over_list, uint32_t *over_count, uint32_t *flags, const uint32_t *chunk_pair);
void launch_slow_caps(hipStream_t st, SetView A, SetView B, const PairDesc *pairs, uint32_t npairs, const uint32_t *list, uint32_t n, uint32_t *cap);
void launch_slow_anchors(hipStream_t st, SetView A, SetView B, const PairDesc *pairs, uint32_t npairs, const uint32_t *list, uint32_t n,
                         const uint32_t *abase, uint32_t *a_qi, uint32_t *a_r, uint32_t *a_rctg, uint32_t *slow_n, uint32_t *flags);
void launch_slow_chain(hipStream_t st, SetView A, SetView B, const PairDesc *pairs, uint32_t npairs, const uint32_t *list, uint32_t n,
                       const uint32_t *abase, const uint32_t *slow_n, const uint32_t *a_qi, const uint32_t *a_r, const uint32_t *a_rctg, int32_t *F,
                       uint32_t *BP, uint64_t *ORD, ChainRec *chains, uint32_t *pair_nch, uint32_t *pair_na, uint32_t *flags);

// ---- finalize (chain_finalize.hip)
void launch_finalize(hipStream_t st, unsigned grid, uint32_t lds_cap, SetView A, SetView B, const PairDesc *pairs, const ChainRec *fast_chains, uint32_t fast_stride,
                     const uint32_t *chunk_state, const ChainRec *chains, const uint32_t *pair_nch, const uint32_t *pair_na, PairOut *out,
                     uint32_t *flags, uint32_t *chunk_mark);
void launch_finalize_global(hipStream_t st, unsigned grid, SetView A, SetView B, const PairDesc *pairs, const ChainRec *fast_chains, uint32_t fast_stride,
                            const uint32_t *chunk_state, const ChainRec *chains, const uint32_t *pair_nch, const uint32_t *pair_na, PairOut *out,
                            uint32_t *flags, uint32_t *chunk_mark, unsigned char *gws, const uint64_t *goff, const uint32_t *glist, const uint32_t *gcap);
void finalize_allow_large_lds();
