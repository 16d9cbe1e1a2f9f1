/*
 * ani_oracle.h -- CPU restatement (TEST INFRASTRUCTURE, not product code) of the all-pairs ANI
 * engine that raufs/skDER reaches through `skani triangle|search|dist|sketch`
 * (/root/reference/src/skDER/skder.py:16-26, 58-61, 103, 119).
 *
 * PARITY STATUS.  The arithmetic lives in the third-party Rust crate `skani`
 * (bluenote-1577/skani; version UNPINNED by the reference: skDER_env.yml:12,
 * bioconda_recipe/meta.yaml:28, Docker/skDER/Dockerfile:10).  Its source is not under
 * /root/reference and it cannot be built or run here (no Rust toolchain, no network).  This
 * file restates skani's published algorithm (Shaw & Yu, Nat. Methods 2023; cited at
 * /root/reference/bin/skder:83-84) and is pinned ONLY against the five golden edge tables the
 * reference's own test run holds (tests/golden/G1..G5: 2-decimal ANI/AF, one species).  Measured
 * residual on G5 (561 pairs): AF rms 0.37 / max 1.1 points, ANI rms 0.14 / max 0.43 points
 * (held out: rms 0.15, oracle/fit_calibration.py; round 5 scored 648 sampling / chaining-rule hypotheses against G5,
 * oracle/sample_hypotheses.py -> profiles/round5_sample_hypotheses.json: none reproduces skani's k-mer sample); the
 * representative listings derived from G1/G5 are reproduced where the goldens are not knife-edge
 * (tests/test_oracle_golden.py).  Beyond those tables: **parity unpinned**.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or load this.
 */
#ifndef ANI_ORACLE_H
#define ANI_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int32_t k, c, marker_k, marker_c;   /* 15, 125, 21, 1000 */
    int32_t min_contig;                 /* 500 */
    int32_t chunk_len;                  /* 20000 */
    int32_t band, bp_band;              /* 50 anchors, 2500 bases */
    int32_t max_gap, max_lin;           /* 300, 5000 */
    int32_t anchor_score, min_anchors;  /* 20, 3 */
    int32_t pad;                        /* 230 */
    int32_t small_pass, rep_floor;      /* 20, 30 */
    int32_t learned;                    /* 1: apply the calibration map (default) */
    /* sampling hypotheses (oracle/sample_hypotheses.py); defaults = include/skder_amd_spec.h */
    int32_t sample_window;              /* 0: keep hash in [0, T); 1: keep hash in [2^63, 2^63 + T) */
    int32_t hash_first_step;            /* 0: ~(key + (key << 21)); 1: ~key + (key << 21) */
    int32_t quarters;                   /* 0: a record as one run; 1, 2: two four-lane models */
    int32_t rule[8];                    /* rule hypotheses of oracle/sample_hypotheses.py; all 0 = the engine's rules */
} oracle_params_t;

void oracle_default_params(oracle_params_t *p);   /* values of include/skder_amd_spec.h */

typedef struct oracle_genome oracle_genome_t;

/* FASTA (plain or gzip) -> sketched genome. NULL + message in err on failure. */
oracle_genome_t *oracle_genome_load(const char *path, const oracle_params_t *p, char *err, size_t errlen);
/* bases: concatenation of n_records FASTA records, lens[i] bytes each (ASCII). */
oracle_genome_t *oracle_genome_from_bases(const uint8_t *bases, const uint32_t *lens, uint32_t n_records,
                                          const char *file_name, const char *first_name,
                                          const oracle_params_t *p);
void oracle_genome_free(oracle_genome_t *g);

uint32_t oracle_genome_n_seeds(const oracle_genome_t *g);
uint32_t oracle_genome_n_markers(const oracle_genome_t *g);
uint32_t oracle_genome_n_contigs(const oracle_genome_t *g);   /* kept records (>= min_contig) */
uint64_t oracle_genome_total_len(const oracle_genome_t *g);   /* sum of kept record lengths */
uint64_t oracle_genome_n50(const oracle_genome_t *g);         /* util.py:686-724 rule, ALL records */
uint32_t oracle_genome_rep_cut(const oracle_genome_t *g);     /* UINT32_MAX = filter off */
const char *oracle_genome_name(const oracle_genome_t *g);     /* header of first kept record */
/* copy-out in position order: canonical k-mer (2 bits/base), gpos = offset of the k-mer's LAST
 * base in the concatenation of kept records, kept-record index, fwd = 1 if the forward strand
 * k-mer was the canonical one.  NULL pointers are skipped. */
void oracle_genome_seeds(const oracle_genome_t *g, uint64_t *kmer, uint32_t *gpos, uint32_t *contig, uint8_t *fwd);
void oracle_genome_markers(const oracle_genome_t *g, uint64_t *markers);     /* sorted unique */
void oracle_genome_contig_offsets(const oracle_genome_t *g, uint32_t *off);  /* n_contigs+1 */

typedef struct {
    int32_t  score;
    uint32_t n_anchors;
    uint32_t n_seeds;         /* seeds of the chunked genome inside [q0, q1] */
    uint32_t q0, q1;          /* gpos of first/last anchor on the chunked genome */
    uint32_t r0, r1;          /* min/max gpos on the other genome */
    uint32_t rctg;            /* kept-record index on the other genome */
    uint32_t kept;            /* 1 if it survived the overlap filter */
    uint32_t chunk;           /* ordinal of its chunk among the chunks that hold anchors */
} oracle_chain_t;

typedef struct {
    int32_t  chunked_query;   /* 1 if the `query` argument was the chunked genome */
    uint32_t n_anchors;
    uint32_t n_chunks;        /* chunks holding at least one anchor */
    uint32_t n_chains_all;    /* chains before the overlap filter */
    uint32_t n_chains;        /* kept chains */
    uint64_t sum_anchors;     /* anchors in kept chains */
    uint64_t sum_seeds;       /* S: chunked-genome seeds inside kept chain spans */
    uint64_t sum_span;        /* sum of (q1 - q0) over kept chains */
    uint64_t aligned_bases;   /* B = sum_span + pad * n_chains */
    uint64_t cell_seeds;      /* N: ALL chunked-genome seeds of the 20 kb cells that hold a kept chain */
    double   ani_raw;         /* cell estimate (sum_anchors / cell_seeds)^(1/k) */
    double   ani_span;        /* span estimate (sum_anchors / sum_seeds)^(1/k) */
    double   ani;             /* after the two-estimate model; fraction */
    double   af_ref;          /* min(1, B / T_ref) */
    double   af_query;        /* min(1, B / T_query) */
} oracle_pair_t;

/* returns 1 if the pair passes the marker screen at `screen_ani_pct` (e.g. 89.5) */
int oracle_screen(const oracle_genome_t *a, const oracle_genome_t *b, double screen_ani_pct,
                  const oracle_params_t *p, uint32_t *shared_out);
/* one pair; ref/query as in skani's Ref/Query columns.  chains (optional): all chains in the
 * order (score desc, q0 asc, r0 asc), with their kept flag; at most chain_cap are written. */
int oracle_pair(const oracle_genome_t *ref, const oracle_genome_t *query, const oracle_params_t *p,
                oracle_pair_t *out, oracle_chain_t *chains, uint32_t chain_cap);

double   oracle_root(uint64_t num, uint64_t den, int k);        /* (num/den)^(1/k), + - * / only */
double   oracle_model_ani(double ani_cell, double ani_span);    /* include/skder_amd_spec.h ANI model */
uint64_t oracle_mm_hash64(uint64_t key);

/* drop-in drivers (listing in, TSV out), mirroring the skani sub-commands skDER spawns */
int oracle_triangle(const char *listing, double min_af_pct, double screen_pct, int threads,
                    const char *out_tsv, const oracle_params_t *p, char *err, size_t errlen);
int oracle_dist(const char *ref_listing, const char *query_listing, double min_af_pct, double screen_pct,
                int threads, const char *out_tsv, const oracle_params_t *p, char *err, size_t errlen);
int oracle_search(const char *listing_db, const char *query_path, double min_af_pct, double screen_pct,
                  int threads, const char *out_tsv, const oracle_params_t *p, char *err, size_t errlen);

#ifdef __cplusplus
}
#endif
#endif
