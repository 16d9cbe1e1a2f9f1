// engine.h -- internal structures of the ANI engine
#pragma once
#include "common.h"

#define IDX_MAX_BUCKET_BITS 15
#define IDX_REP_HIST ANI_REP_HIST
#define HIT_POS_BITS 24        /* hit words: position in the low 24 bits, record index mod 64 in bits 24..29 */
#define HIT_POS_MASK 0x00FFFFFFu
#define HIT_KEY_MASK 0xBF000000u   /* strand (bit 31) and record tag (bits 24..29) */

// per-genome record on the device (index stage)
struct GenomeMeta {
    uint64_t seed_off;      // offset of the genome's seeds in the seed arrays
    uint64_t bucket_off;    // offset of its bucket-offset table (nb+1 entries)
    uint64_t marker_off;
    uint64_t total_len;     // sum of kept record lengths
    uint64_t rec_goff_off;  // offset into d_rec_goff (n_rec+1 entries)
    uint64_t chunk_off;     // offset of its chunk_start table (n_chunks+1 entries)
    uint32_t n_seeds;
    uint32_t n_markers;
    uint32_t n_rec;
    uint32_t bucket_bits;
    uint32_t n_chunks;      // filled by the index kernel
    uint32_t rep_cut;       // filled by the index kernel
};

// marker -> genome-list inverted index of a sketch set (screen.hip); built on the first screen against
// the set and kept, so the per-representative searches of low_mem_greedy do not rebuild it
struct ScreenIndex {
    bool built = false;
    uint64_t ts = 0;                       // table size (power of two)
    DevBuf<uint64_t> keys;                 // open-addressing table of marker k-mers
    DevBuf<uint32_t> loff, list, slot_of;  // per slot: offset of its genome list; lists; slot of every marker occurrence
};

struct skder_sketches {
    skder_ctx *ctx = nullptr;
    uint32_t n_genomes = 0;
    bool indexed = false;
    bool index_pending = false;            // index kernels enqueued (index_begin), results not fetched yet
    DevBuf<uint32_t> idx_list;             // genome lists of the index kernels, alive until index_finish
    DevBuf<uint4> idx_packed;              // (k-mer, position, record, -) per seed: one gather instead of three, index build only
    std::vector<uint32_t> idx_small, idx_big;
    std::vector<uint8_t> full_index;        // per genome: bucket index built here (else chunk tables only: another GPU owns it)
    uint32_t partial_index = 0;             // genomes that have chunk tables only (0: every genome can be probed)
    hipStream_t idx_stream = nullptr;
    // raw sketches
    DevBuf<uint32_t> seed_kmer, seed_gpos, seed_ctg;   // position order
    DevBuf<uint64_t> markers;                          // sorted unique per genome
    std::vector<uint64_t> h_seed_off{0}, h_marker_off{0}, h_genome_len;
    std::vector<uint32_t> h_genome_nrec, h_rec_goff;
    // index
    DevBuf<GenomeMeta> d_meta;
    std::vector<GenomeMeta> h_meta;
    DevBuf<uint32_t> d_rec_goff;
    DevBuf<uint32_t> skmer, sgpos, sctg;   // by-(kmer,gpos) order inside each hash bucket
    DevBuf<uint32_t> stag;                 // sgpos | (sctg & 63) << 24: the word the join hands to the chaining kernel
    DevBuf<uint32_t> boff;                 // bucket offset tables
    DevBuf<uint32_t> pchunk;               // chunk id of every seed (position order)
    DevBuf<uint8_t> pcs;                   // 1 where a seed is the first of its chunk (position order): the run extraction reads 4 flags per lane
    DevBuf<uint32_t> chunk_start;          // first seed of every chunk (+ end sentinel), per genome
    ScreenIndex screen;
};

void sketch_batch_impl(skder_sketches *s, const uint8_t *d_bases, const skder_batch_t *b);
void index_impl(skder_sketches *s);                       // build the index and wait for it
void index_begin(skder_sketches *s, hipStream_t st, const uint8_t *full = nullptr);   // enqueue the index build on st (after what ctx->stream
                                                                                      // holds now); full[g] == 0: chunk tables only for genome g
void index_promote(skder_sketches *s, const std::vector<uint32_t> &genomes);          // full index for some of those afterwards
void index_set_rep_cuts(skder_sketches *s, const uint32_t *in, const uint8_t *mask);
void index_finish(skder_sketches *s);                    // wait for it, fetch the per-genome results
void triangle_rows_impl(skder_sketches *s, uint32_t row_begin, uint32_t row_stride, double screen_pct);
void rectangle_impl(skder_sketches *refs, skder_sketches *queries, double screen_pct, const uint8_t *live_refs = nullptr);   // live_refs: per genome of refs, 0 = leave its pairs out
void screen_rows_impl(skder_sketches *s, uint32_t row_begin, uint32_t row_stride, double screen_pct,
                      std::vector<uint32_t> &pref, std::vector<uint32_t> &pquery);
void pairs_probed_impl(skder_sketches *SA, skder_sketches *SB, const uint32_t *ref, const uint32_t *query, uint64_t n, uint32_t *probed,
                       uint8_t *probed_is_query);
void chain_pairs_impl(skder_sketches *SA, skder_sketches *SB, const uint32_t *ref, const uint32_t *query, uint64_t n);
void synth_fill_impl(skder_ctx *ctx, uint8_t *d_bases, const skder_batch_t *b, const uint64_t *lineage,
                     const uint32_t *params);

// Multiplication by an odd constant is a bijection of the 30-bit k-mers: the bucket is the top `bits` bits
// of the 30-bit product and the remaining 30 - bits bits (kmer_rem) identify the k-mer inside its bucket,
// so an index can keep 16-bit remainders instead of k-mers once bits >= 14.
__host__ __device__ inline uint32_t kmer_mix(uint32_t kmer) { return (kmer * 0x9E3779B1u) & 0x3FFFFFFFu; }
__host__ __device__ inline uint32_t kmer_bucket(uint32_t kmer, uint32_t bits) { return kmer_mix(kmer) >> (30u - bits); }
__host__ __device__ inline uint32_t kmer_rem(uint32_t kmer, uint32_t bits) { return kmer_mix(kmer) & ((1u << (30u - bits)) - 1u); }
