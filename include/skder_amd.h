/*
 * skder_amd.h -- C ABI of libskder_amd.so, the MI355X (gfx950) all-pairs ANI engine that stands in
 * for the `skani` sub-processes raufs/skDER spawns.  Plain pointers and sizes only; no torch types.
 *
 * The reference has no FFI for this path: it builds shell command lines and checks that the output
 * file exists (/root/reference/src/skDER/util.py:636-652).  Section A below therefore mirrors those
 * command lines one entry point per sub-command (file paths in, skani-format TSV out, "no output
 * file" == failure).  Section B is the device-level interface the same entry points are built from;
 * bench.py, the parity tests and the multi-GPU driver (skder_amd/multigpu.py) call it with device
 * pointers they own (torch is used there only as an allocator / RCCL front end).
 *
 * NUMERICAL STATUS.  Formats, row order, names and filter rules are skani's as the
 * reference's golden tables show them; the ANI / AF VALUES are not: skani's learned-ANI model and k-mer sample could not
 * be reconstructed, so ANI is skani's published chunk-level estimator plus a two-parameter stand-in (skder_amd_spec.h).
 * Against the reference's golden table (561 pairs, one species, ANI 96.4-100): ANI rms 0.14 / max 0.42 points, AF rms
 * 0.37 / max 1.10; unpinned beyond.  skder_edge_t carries ani_raw and its integer counts for callers with their own model.
 * (648 sampling and chaining-rule hypotheses were scored against that table in round 5, profiles/round5_sample_hypotheses.json:
 * none reproduces skani's sample; the residual is what an independent FracMinHash sample leaves.)
 *
 * Every function returns 0 on success; on failure it returns non-zero and writes a message into
 * err[0..errlen) -- the Python shim raises RuntimeError from it, as util.runCmd does (util.py:652).
 * There is NO CPU fallback anywhere behind this header: without a gfx950 device every compute entry
 * point fails with an error.
 *
 * ENVIRONMENT SWITCHES read by the library (all fourteen of them; none is needed for normal use; results are identical under every one):
 *   SKDER_AMD_DEBUG=1|2        per-batch counters (chunks per path, decline causes) and host timings on stderr; 2: one line per ingested file
 *   SKDER_AMD_QUEUES=n         HIP queues the chaining batches alternate between (default 2; 1 = batch after batch: per-kernel timings)
 *   SKDER_AMD_CHUNK_BUDGET=n   chunks per chaining batch (default 6 M);  SKDER_AMD_PAIR_BUDGET=n  candidate pairs per screening block (2^31);
 *   SKDER_AMD_REC_DIV=n        seeds per run-record slot (default 4)                       -- the three are exercised by the parity tests
 *   SKDER_AMD_FORCE_SLOW=1     every chunk through the general (unabridged) chaining kernel;  SKDER_AMD_NO_SIEVE=1  none settled by the sieve;
 *   SKDER_AMD_NO_ROWS=1        declined chunks to the one-wavefront-per-chunk kernel instead of the rows kernel  -- parity A/B of the chaining paths
 *   SKDER_AMD_RUNS_REFILL=n    free lanes of a wavefront at which the run loop writes finished chunks out and hands new ones over (default 24; 1 - 64)
 *   SKDER_AMD_IO_THREADS=n     reader threads of the ingest (default: the cgroup's CPUs, at most 128);  SKDER_AMD_IO_BATCH_MB=n  pinned staging batch (256)
 *   SKDER_AMD_IO_TWO_PHASE=1   .gz files through memory of their own instead of straight into the staging buffer
 *   SKDER_AMD_HOST_PARSE=1     FASTA parsed by the host reader instead of the device;  SKDER_AMD_FASTA_WAVE=1  by the one-wavefront-per-file kernel
 * Read by the Python host mirror (skder_amd/skder.py): SKDER_AMD_DEVICE, SKDER_AMD_DEVICES (device list of the drop-in entry points),
 * SKDER_AMD_SEARCH_BATCH, SKDER_AMD_SEARCH_ALL (lowMemGreedyDerep's speculative batches); by bench.py: SKDER_AMD_FORCE_DIST, SKDER_AMD_DIST_BACKEND,
 * SKDER_AMD_EXCHANGE=components (N > 1: markers all-gathered, seeds to the owner of each connected component; skder_amd/multigpu.py),
 * SKDER_AMD_OTHER_EXCHANGE=1 (N > 1: three extra steps of the exchange that was not selected, outside the timed region).
 */
#ifndef SKDER_AMD_H
#define SKDER_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ======================================================================================
 * A. drop-in entry points (one per skani sub-command used by the reference)
 * ====================================================================================== */

/* `skani triangle -l LISTING --min-af MIN_AF -E -s SCREEN -t T -o OUT`
 * replaces /root/reference/src/skDER/skder.py:16-26 (runSkaniTriangle).
 * Writes the 7-column edge table (header, %.2f percentages, first-record>=500bp names, rows kept
 * iff max(AF_ref,AF_query) >= min_af on unrounded values, skani's row order).  The file is written
 * to a temporary name and renamed, so it exists only if the call succeeded. */
int skder_amd_triangle(const char *listing, double min_af_pct, double screen_pct, int device,
                       const char *out_tsv, char *err, size_t errlen);

/* `skani dist --rl REFS --ql QUERIES [-s SCREEN] -o OUT`
 * replaces skder.py:58-61 (runSkaniDist) and cidder.py:362-364.  Rows grouped by query in --ql
 * order, references by ANI descending; skani's dist defaults are min_af 15, screen 80. */
int skder_amd_dist(const char *ref_listing, const char *query_listing, double min_af_pct,
                   double screen_pct, int device, const char *out_tsv, char *err, size_t errlen);

/* `skani sketch -l LISTING -o DB` replaces skder.py:103-104: the sketch database stays resident in
 * HBM behind an opaque handle instead of a directory on disk. NULL on failure. */
typedef struct skder_db skder_db_t;
skder_db_t *skder_amd_sketch(const char *listing, int device, char *err, size_t errlen);
/* `skani search QUERY -d DB -o OUT` replaces skder.py:119-120 (one call per representative).
 * QUERY may be a path of the listing the DB was built from (its sketch is reused) or any FASTA.
 * Same 7 columns as triangle; includes the self hit; Ref_file strings equal the listing lines
 * byte for byte (they are compared with N50-table keys at skder.py:117,129). */
int skder_amd_search(skder_db_t *db, const char *query_path, double min_af_pct, double screen_pct,
                     const char *out_tsv, char *err, size_t errlen);
void skder_amd_db_free(skder_db_t *db);

/* The same over SEVERAL GPUs of the node from one process (`skani ... -t T` means "use the whole machine",
 * skder.py:18): `devices` lists n_devices GPU indices.  GPU d reads and sketches the d-th share of the listing; the raw
 * sketches are exchanged between the GPUs (peer copies over xGMI); GPU d builds the k-mer index of the genomes it owns
 * (listing index mod n_devices), screens every n_devices-th row and chains the pairs that probe one of its genomes.
 * The table and its row order are those of the one-GPU call.  A database handle from skder_amd_sketch_multi serves
 * skder_amd_search / _search_batch / _db_triangle the same way (queries are chained on the GPU that owns the database
 * genome of a pair); skder_amd_db_save writes the same store as a one-GPU database. */
int skder_amd_triangle_multi(const char *listing, double min_af_pct, double screen_pct, const int *devices, int n_devices,
                             const char *out_tsv, const char *n50_tsv /* may be NULL */, char *err, size_t errlen);
skder_db_t *skder_amd_sketch_multi(const char *listing, const int *devices, int n_devices, const char *n50_tsv /* may be NULL */,
                                   char *err, size_t errlen);

/* The `-p` free-text skani parameter string (bin/skder:132,199-201): accepts "-s <float>" only and
 * rejects any other skani flag loudly.  On success *screen_pct is set (unchanged if no -s). */
int skder_amd_parse_skani_params(const char *params, double *screen_pct, char *err, size_t errlen);

/* ======================================================================================
 * B. device-level interface
 * ====================================================================================== */

typedef struct skder_ctx skder_ctx_t;
skder_ctx_t *skder_amd_ctx_create(int device, char *err, size_t errlen);
void skder_amd_ctx_destroy(skder_ctx_t *ctx);
/* stream all kernels of this context are launched on (a hipStream_t) */
void *skder_amd_ctx_stream(skder_ctx_t *ctx);
const char *skder_amd_last_error(skder_ctx_t *ctx);

/* Layout of a batch of genomes whose bases are resident in HBM (host arrays, copied by the call):
 * record r (a kept FASTA record, >= 500 bp) occupies d_bases[rec_off[r] .. rec_off[r]+rec_len[r]),
 * ASCII, rec_off[r] a multiple of 32, with >= 32 readable bytes in front of the first record and
 * >= SKDER_TILE+32 readable bytes behind the last; genome g owns records
 * [genome_rec_begin[g], genome_rec_begin[g+1]). */
#define SKDER_TILE 8192
typedef struct {
    uint32_t n_genomes;
    uint32_t n_records;
    const uint64_t *rec_off;
    const uint32_t *rec_len;
    const uint32_t *genome_rec_begin;   /* n_genomes + 1 */
} skder_batch_t;

/* A sketch set: FracMinHash seeds + markers of a list of genomes, resident in HBM. */
typedef struct skder_sketches skder_sketches_t;
skder_sketches_t *skder_amd_sketches_new(skder_ctx_t *ctx);
void skder_amd_sketches_free(skder_sketches_t *s);
/* HOT KERNEL: sketch `batch` (bases already on the device) and append the genomes to `s`. */
int skder_amd_sketch_batch(skder_sketches_t *s, const uint8_t *d_bases, const skder_batch_t *batch);
/* capacity hint (expected totals over all batches): the seed and marker arrays are allocated once
 * instead of growing batch by batch.  Purely an optimisation; smaller or larger totals still work. */
int skder_amd_sketches_reserve(skder_sketches_t *s, uint64_t n_seeds, uint64_t n_markers);
/* sizes / raw device arrays (position-ordered seeds; sorted unique markers), for all-gather:
 *   seed_off[n_genomes+1], seed_kmer/gpos/ctg[n_seeds]; marker_off[n_genomes+1], markers[n_markers];
 *   genome_len[n_genomes] (sum of kept record lengths), genome_nrec[n_genomes],
 *   rec_goff: per genome record start offsets concatenated (genome_nrec[g]+1 entries each). */
typedef struct {
    uint32_t n_genomes;
    uint64_t n_seeds, n_markers, n_rec_goff;
    const uint32_t *d_seed_kmer;   /* canonical 15-mer | fwd<<31 */
    const uint32_t *d_seed_gpos;
    const uint32_t *d_seed_ctg;
    const uint64_t *d_markers;
    const uint64_t *h_seed_off;    /* host */
    const uint64_t *h_marker_off;  /* host */
    const uint64_t *h_genome_len;  /* host */
    const uint32_t *h_genome_nrec; /* host */
    const uint32_t *h_rec_goff;    /* host */
} skder_raw_view_t;
int skder_amd_sketches_view(skder_sketches_t *s, skder_raw_view_t *out);
/* append genomes from raw arrays (device pointers for seeds/markers, host pointers for the rest);
 * d_seed_ctg may be NULL: the record indices are then derived from d_seed_gpos and h_rec_goff */
int skder_amd_sketches_append_raw(skder_sketches_t *s, const skder_raw_view_t *raw);
/* build the per-genome lookup structures (k-mer bucket index, chunk tables, repetitive cut-offs) */
int skder_amd_sketches_index(skder_sketches_t *s);

/* one reported genome pair */
typedef struct {
    uint32_t ref;            /* genome index on the reference side */
    uint32_t query;          /* genome index on the query side */
    double ani;              /* fraction, after the ANI model of skder_amd_spec.h (what the TSV prints) */
    double af_ref;
    double af_query;
    uint32_t n_chains;       /* kept chains */
    uint32_t n_anchors;
    uint64_t aligned_bases;
    uint64_t sum_anchors;    /* A: anchors in the kept chains */
    uint64_t sum_seeds;      /* S: seeds of the chunked genome inside the kept chains' spans */
    uint64_t cell_seeds;     /* N: all seeds of the chunked genome in the 20 kb cells that hold a kept chain */
    double ani_raw;          /* (A/N)^(1/15): the chunk-level k-mer estimate BEFORE the model, for callers with a
                              * model of their own ((A/S)^(1/15), the span estimate, follows from the counts) */
} skder_edge_t;

/* Upper triangle rows i = row_begin, row_begin+row_stride, ... of `s` against all j > i:
 * marker screen at screen_pct, anchors, chaining, ANI/AF.  Edges (every screened pair with at least
 * one chain) are appended to a host buffer owned by the context; the edges pointer and n_edges are valid until the
 * next call on the same context. */
int skder_amd_triangle_rows(skder_sketches_t *s, uint32_t row_begin, uint32_t row_stride,
                            double screen_pct, const skder_edge_t **edges, uint64_t *n_edges);
/* rectangle: every genome of `refs` against every genome of `queries` (dist / search) */
int skder_amd_rectangle(skder_sketches_t *refs, skder_sketches_t *queries, double screen_pct,
                        const skder_edge_t **edges, uint64_t *n_edges);

/* ---- one pair matrix over several GPUs.  Every GPU holds the raw sketches of ALL genomes (all-gather),
 * but builds the k-mer bucket index only for the genomes it OWNS (full[g] != 0; the others get their chunk tables only)
 * and chains the pairs that PROBE one of its genomes; the marker screen is dealt out by rows.  skder_amd/multigpu.py
 * (one process per GPU, RCCL) and skder_amd_triangle_multi below (one process, several GPUs) are built from these. */
/* The build is enqueued on the context's second queue and returns: skder_amd_screen_rows, which reads the markers only, runs
 * beside it; every call that needs the index or its per-genome results (_rep_cuts, _pairs_probed, _chain_pairs) waits for it. */
int skder_amd_sketches_index_part(skder_sketches_t *s, const uint8_t *full /* n_genomes flags; NULL: all */);
/* repetitive-k-mer cut-off of every genome: what this set computed for the genomes it indexed fully, 0xFFFFFFFF ("unknown
 * here") for the others; the element-wise minimum over the GPUs is the table to install with _set_rep_cuts */
int skder_amd_sketches_rep_cuts(skder_sketches_t *s, uint32_t *out /* n_genomes */);
int skder_amd_sketches_set_rep_cuts(skder_sketches_t *s, const uint32_t *in /* n_genomes */);
/* candidate pairs (marker screen) of triangle rows row_begin, row_begin + row_stride, ...: host arrays owned by the
 * context, valid until its next call */
int skder_amd_screen_rows(skder_sketches_t *s, uint32_t row_begin, uint32_t row_stride, double screen_pct,
                          const uint32_t **ref, const uint32_t **query, uint64_t *n_pairs);
/* the genome each pair probes (the other one is cut into 20 kb chunks): its index, and whether it is the pair's query */
int skder_amd_pairs_probed(skder_sketches_t *refs, skder_sketches_t *queries, const uint32_t *ref, const uint32_t *query,
                           uint64_t n_pairs, uint32_t *probed, uint8_t *probed_is_query /* may be NULL */);
/* anchors, chaining, ANI / AF of an explicit list of pairs (ref in `refs`, query in `queries`; the same set twice for a
 * triangle).  A probed genome that has only chunk tables here gets its index built first. */
int skder_amd_chain_pairs(skder_sketches_t *refs, skder_sketches_t *queries, const uint32_t *ref, const uint32_t *query,
                          uint64_t n_pairs, const skder_edge_t **edges, uint64_t *n_edges);

/* device-to-device copy on the context's stream, complete on return (lets a caller that holds device
 * memory of its own -- e.g. a torch tensor for the RCCL exchange -- take a copy of the raw arrays) */
int skder_amd_copy_d2d(skder_ctx_t *ctx, void *dst, const void *src, size_t bytes);

/* timing of the last triangle_rows/rectangle/sketch_batch call, milliseconds by HIP events on the
 * context's stream: [0] sketch kernel, [1] sketch post-processing + index, [2] screen,
 * [3] chaining fast path, [4] chaining slow path, [5] finalize; counts: [6] pairs screened in, [7] anchors */
int skder_amd_last_timing(skder_ctx_t *ctx, double *out8);
/* device milliseconds of the last seed-index build (skder_amd_sketches_index, or the build that
 * triangle_rows / rectangle start themselves for a set that is not indexed yet: it then runs beside
 * the marker screen on a second stream) */
double skder_amd_last_index_ms(skder_ctx_t *ctx);
/* device milliseconds of the run-extraction kernel (hit words -> run records) in the last triangle_rows / rectangle call */
double skder_amd_last_runs_ms(skder_ctx_t *ctx);
/* counters of the last triangle_rows/rectangle call: [0] chunks processed, [1] chunks that needed the
 * unabridged (slow) chaining path */
int skder_amd_last_counters(skder_ctx_t *ctx, uint64_t *out4);
/* Memory the library keeps between calls on a device -- the ingest's pinned staging buffers with their device copies (about 0.6 GB of
 * pinned host memory and 1.2 GB of HBM at the default batch size), the device allocator's cached blocks and the host list the table
 * writers keep for the next small table (at most 256 MB) -- handed back.  For long-lived host applications; 0: released, 1: the staging set is in use by a running call (the rest was released). */
int skder_amd_release_cached_buffers(int device);
/* multi-GPU calls of one process (skder_amd_triangle_multi, skder_amd_sketch_multi): ordered device pairs found WITHOUT peer access so
 * far -- their copies are staged through host memory by the runtime (correct, an order of magnitude slower than xGMI).  0 on a healthy node. */
uint32_t skder_amd_peer_fallbacks(void);
/* Which ANI the edge records, and every table written from them, carry (process-wide; returns the previous setting, -1 for an invalid
 * argument): 0 = after the learned-ANI stand-in of skder_amd_spec.h (default; what `skani` prints by default), 1 = the raw chunk-level
 * k-mer estimate (A/N)^(1/15), the counterpart of skani's `--no-learned-ani`.  The stand-in is fitted to skani's output on real genomes,
 * whose changes cluster; on simulated genomes with independent substitutions it reads 1.24 x the true divergence and the raw estimate is
 * the unbiased one.  skder_edge_t.ani_raw holds the raw estimate under either setting. */
int skder_amd_set_ani_output(int raw);

/* synthetic genomes generated ON the device (SURVEY 8d recipe; bench.py / tests):
 * fills d_bases for one batch from (seed, species, strain, isolate) lineage ids. See synth.h. */
int skder_amd_synth_fill(skder_ctx_t *ctx, uint8_t *d_bases, const skder_batch_t *batch,
                         const uint64_t *genome_lineage /* 3 per genome: species,strain,isolate seeds */,
                         const uint32_t *genome_params /* 4 per genome, see synth.h */);

/* descendants of REAL assemblies generated ON the device (descend.hip; bench.py / tests): a descendant keeps its parent's records;
 * per record at most one structural event (three per genome), then short indels and substitutions at the given rates, everything a
 * pure function of (seed, record, position).  Two calls, no state between them: _lengths computes every record's length (the caller
 * lays the batch out: records of 500 bases and more, 32-byte aligned), _fill writes the bases.  rec_len_out / rec_out_off: one entry
 * per record of every descendant's parent, in order; rec_out_off = ~0 drops a record. */
typedef struct {
    uint32_t parent;               /* genome index in the ancestors' batch */
    uint32_t sub_ppm, indel_ppm;   /* substitutions / short-indel events per million positions */
    uint32_t n_events;             /* structural events, 0..3, no two in one record */
    uint64_t seed;
    struct { uint32_t rec, type, s, n, b; } ev[3];   /* rec: record inside the parent; type 0 inversion of [s, s+n), 1 [s, s+n) moved to
                                                        position b of the record without it, 2 deletion of [s, s+n) */
    uint32_t pad;
} skder_descendant_t;
int skder_amd_descend_lengths(skder_ctx_t *ctx, const uint8_t *d_anc_bases, const skder_batch_t *anc, const skder_descendant_t *desc,
                              uint32_t n_desc, uint32_t *rec_len_out, uint32_t n_rec_out);
int skder_amd_descend_fill(skder_ctx_t *ctx, const uint8_t *d_anc_bases, const skder_batch_t *anc, const skder_descendant_t *desc,
                           uint32_t n_desc, uint8_t *d_out_bases, const uint64_t *rec_out_off, uint32_t n_rec_out);

/* ======================================================================================
 * C. the callers either side of the path: all work on the resident database
 * ====================================================================================== */

/* 8f-2  N50 fused into ingest.  The pass that parses/uploads a listing's FASTA files also yields
 * the N50 of every file exactly as util.n50_calc does (/root/reference/src/skDER/util.py:686-724:
 * all records incl. < 500 bp, half = int(sum/2), first cumulative >= half, lengths descending).
 * n50_tsv (may be NULL) receives `path<TAB>N50` lines in LISTING order, the format of
 * Concatenated_N50.txt (util.py:476-501, bin/skder:322-323). */
int skder_amd_triangle_n50(const char *listing, double min_af_pct, double screen_pct, int device,
                           const char *out_tsv, const char *n50_tsv, char *err, size_t errlen);
skder_db_t *skder_amd_sketch_n50(const char *listing, int device, const char *n50_tsv, char *err, size_t errlen);
/* genomes in the database, their listing lines and N50 values (listing order) */
uint32_t skder_amd_db_size(skder_db_t *db);
const char *skder_amd_db_path(skder_db_t *db, uint32_t i);
uint64_t skder_amd_db_n50(skder_db_t *db, uint32_t i);

/* A database from a sketch set that is ALREADY in HBM (built with the device-level interface of section B, e.g. by a
 * caller that produced the bases on the device or received the sketches from another GPU): the raw sketches are copied
 * (device to device) into a database of their own on `device` and indexed; `s` stays the caller's.  paths / first_names:
 * n strings each (the Ref_file / Ref_name columns; first_names may be NULL: empty names), n50 may be NULL (zeros).
 * A query path that equals one of `paths` is served from the resident sketch, as with skder_amd_sketch. */
skder_db_t *skder_amd_db_from_sketches(skder_sketches_t *s, int device, uint32_t n /* entries of the three arrays: must equal the set's genome count */,
                                       const char *const *paths, const char *const *first_names, const uint64_t *n50, char *err, size_t errlen);

/* 8f-1  edge list handed over IN MEMORY.  All-pairs table of the resident database: the rows of
 * `skani triangle` (Ref = the genome whose path sorts first, skani's row order, --min-af applied),
 * `ref`/`query` being LISTING indices.  out_tsv may be NULL (no text round trip: the selection step
 * reads *edges); the array is owned by the database and valid until its next call. */
int skder_amd_db_triangle(skder_db_t *db, double min_af_pct, double screen_pct, const char *out_tsv,
                          const skder_edge_t **edges, uint64_t *n_edges, char *err, size_t errlen);

/* 8f-1  representative selection ON the edge rows (host code, select.cpp): the counterparts of skDERsum + `sort -k 2 -gr` + the
 * greedy loop (/root/reference/src/skDER/skDERsum.cpp:60-165, skder.py:136-165), of skDERcore (skDERcore.cpp:60-224, the rule the
 * code implements) and of determineClusters (skder.py:168-277), byte-identical with the files the reference writes from the same
 * table.  rows: ref/query = indices into paths (the array skder_amd_db_triangle hands over, in the table's row order -- member
 * lists and the clustering table follow it); values are rounded to the table's two decimals here.  n50: listing order.
 * display_names (may be NULL = paths): the names written into skDER_Results.txt / skDER_Clustering.txt, i.e. the reference's
 * mge_proc_to_unproc_mapping (skder.py:76-92, 160-163, 236-253).  Output file names may be NULL (not written); reps (room for
 * n_genomes) receives the representatives in the order of skDER_Results.txt. */
int skder_amd_select_greedy(const skder_edge_t *rows, uint64_t n_rows, uint32_t n_genomes, const char *const *paths, const uint64_t *n50,
                            const char *const *display_names, double min_ani_pct, double min_af_pct,
                            const char *info_txt /* Genome_Information_for_Greedy_Clustering.txt */, const char *sorted_txt /* ....sorted.txt */,
                            const char *results_txt /* skDER_Results.txt */, uint32_t *reps, uint32_t *n_reps, char *err, size_t errlen);
int skder_amd_select_dynamic(const skder_edge_t *rows, uint64_t n_rows, uint32_t n_genomes, const char *const *paths, const uint64_t *n50,
                             const char *const *display_names, double min_ani_pct, double min_af_pct, double max_af_diff_pct,
                             const char *results_txt, uint32_t *reps, uint32_t *n_reps, char *err, size_t errlen);
int skder_amd_select_clusters(const skder_edge_t *rows, uint64_t n_rows, uint32_t n_genomes, const char *const *paths,
                              const char *const *display_names, const uint32_t *reps, uint32_t n_reps, double af_cutoff_pct,
                              double ani_cutoff_pct, const char *clustering_txt /* skDER_Clustering.txt */, char *err, size_t errlen);
/* rows whose table TEXT passes: ANI >= ani_cut and column af_column (4: Align_fraction_ref, 5: Align_fraction_query) >= af_cut -- the test
 * of skder.py:127-129 on a `skani search` table, on edge records; pass[i] = 0 / 1 */
int skder_amd_rows_pass(const skder_edge_t *rows, uint64_t n_rows, double ani_cut_pct, double af_cut_pct, int af_column, uint8_t *pass);
/* hundredths of a percent as `%.2f` prints (double)((float)fraction * 100.0f): the rounding the table writer and the selection share */
int64_t skder_amd_pct2_cents(float fraction);

/* 8f-3  speculative batch of `skani search` calls (skder.py:116-133 evaluates one representative
 * at a time): the rows of n_queries candidates are computed in one pass.  Rows come grouped by
 * query in the order given, references by ANI descending, `query` = index into query_paths.
 * out_tsvs may be NULL, or hold one output name per query (NULL entries are skipped). */
int skder_amd_search_batch(skder_db_t *db, const char *const *query_paths, uint32_t n_queries, double min_af_pct,
                           double screen_pct, const char *const *out_tsvs, const skder_edge_t **edges,
                           uint64_t *n_edges, char *err, size_t errlen);
/* The same with the database genomes a caller still cares about: live[i] != 0 (skder_amd_db_size entries; NULL: all).  Candidate pairs
 * of the other genomes are screened but not chained and have no rows.  lowMemGreedyDerep (skder.py:116-133) only ever ADDS the Ref of a
 * row to its set of accounted genomes, so rows of genomes that are accounted for already, or were representatives earlier in the order,
 * cannot change its result: with those masked out the listing is the same and the chaining shrinks with the live set. */
int skder_amd_search_batch_live(skder_db_t *db, const char *const *query_paths, uint32_t n_queries, double min_af_pct,
                                double screen_pct, const char *const *out_tsvs, const uint8_t *live, const skder_edge_t **edges,
                                uint64_t *n_edges, char *err, size_t errlen);

/* 8f-4  sketch store on disk (the counterpart of the directory `skani sketch -o` creates,
 * skder.py:102-104): one file holding the raw sketches, record tables, names, paths and N50s of a
 * database, so a run over 10^4..10^5 genomes can resume without re-ingesting FASTA.  The format is
 * this library's own (little endian, versioned magic); a file that does not match is refused. */
int skder_amd_db_save(skder_db_t *db, const char *store_path, char *err, size_t errlen);
skder_db_t *skder_amd_db_load(const char *store_path, int device, char *err, size_t errlen);

#ifdef __cplusplus
}
#endif
#endif
