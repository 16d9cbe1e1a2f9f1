/*
 * ani_oracle.c -- CPU restatement of the skani-style ANI engine behind skDER's hot path.
 * TEST INFRASTRUCTURE ONLY (see ani_oracle.h for the parity statement: "parity unpinned" beyond
 * the golden tables G1..G5).  Plain C99 + zlib + OpenMP.
 *
 * What is restated, and from where:
 *   - call sites / boundary semantics: /root/reference/src/skDER/skder.py:10-63, 95-134 and
 *     /root/reference/src/skDER/util.py:636-652 (output file existence is the only success test);
 *   - observable skani behaviour verified on the reference's golden tables: SURVEY.md 8(c) V1..V10
 *     (header/columns, %.2f, row order, first-record>=500 names, max(AF) filter on unrounded
 *     values, one aligned-base count per pair, AF cap at 1, records < 500 bp ignored);
 *   - the algorithm itself (FracMinHash seeds c=125 k=15, markers c=1000 k=21, 20 kb chunks,
 *     banded anchor chaining, containment ANI, aligned fraction): skani's published description
 *     (Shaw & Yu 2023) -- third-party dependency `skani`, version unpinned, source absent.
 *     Choices the description leaves open (hash first step, which genome is chunked, overlap
 *     filter, padding) were selected by residual against golden table G5; DESIGN.md lists them.
 * Everything numeric that the HIP path must reproduce is defined with integers or with
 * + - * / on doubles in a fixed order (no libm, -ffp-contract=off), so "parity with the oracle"
 * means bit-equal.
 */
#define _GNU_SOURCE
#include "ani_oracle.h"
#include "../include/skder_amd_spec.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <zlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ params */

void oracle_default_params(oracle_params_t *p)
{
    memset(p, 0, sizeof(*p));
    p->k = ANI_K;
    p->c = ANI_C;
    p->marker_k = ANI_MARKER_K;
    p->marker_c = ANI_MARKER_C;
    p->min_contig = ANI_MIN_CONTIG;
    p->chunk_len = ANI_CHUNK_LEN;
    p->band = ANI_BAND;
    p->bp_band = ANI_BP_BAND;
    p->max_gap = ANI_MAX_GAP;
    p->max_lin = ANI_MAX_LIN;
    p->anchor_score = ANI_ANCHOR_SCORE;
    p->min_anchors = ANI_MIN_ANCHORS;
    p->pad = ANI_PAD;
    p->small_pass = ANI_SMALL_PASS;
    p->rep_floor = ANI_REP_FLOOR;
    p->learned = 1;
    p->sample_window = ANI_SAMPLE_WINDOW;
    p->hash_first_step = 0;
    p->quarters = 0;
}

/* ------------------------------------------------------------------ hashing */

/* minimap2's invertible 64-bit mix (Thomas Wang), which skani uses for FracMinHash sampling
 * (SURVEY R1), with the first step as skani's Rust source spells it:
 * `key = !key.wrapping_add(key << 21)` == ~(key + (key << 21)) -- see include/skder_amd_spec.h. */
static inline uint64_t mm_hash64_v(uint64_t key, int first_step)
{
    key = first_step ? ~key + (key << 21) : ~(key + (key << 21));
    key = key ^ (key >> 24);
    key = (key + (key << 3)) + (key << 8);
    key = key ^ (key >> 14);
    key = (key + (key << 2)) + (key << 4);
    key = key ^ (key >> 28);
    key = key + (key << 31);
    return key;
}
uint64_t oracle_mm_hash64(uint64_t key)
{
    key = ~(key + (key << 21));
    key = key ^ (key >> 24);
    key = (key + (key << 3)) + (key << 8);
    key = key ^ (key >> 14);
    key = (key + (key << 2)) + (key << 4);
    key = key ^ (key >> 28);
    key = key + (key << 31);
    return key;
}

static inline uint32_t base_code(uint8_t b)
{
    switch (b) {
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 0; /* A and everything else */
    }
}

/* ------------------------------------------------------------------ genome */

struct oracle_genome {
    char *file_name;
    char *first_name;      /* header line of the first kept record */
    uint32_t n_contigs;    /* kept */
    uint32_t *ctg_off;     /* n_contigs+1, offsets into the concatenation of kept records */
    uint64_t total_len;    /* sum kept */
    uint64_t n50;          /* over ALL records */
    uint32_t n_seeds;
    uint64_t *s_kmer;      /* position order */
    uint32_t *s_gpos;
    uint32_t *s_ctg;
    uint8_t *s_fwd;
    uint32_t *by_kmer;     /* permutation of 0..n_seeds-1 sorted by (kmer, index) */
    uint32_t rep_cut;      /* k-mers occurring more often than this are not anchored */
    uint32_t n_markers;
    uint64_t *markers;     /* sorted unique */
};

typedef struct { uint64_t *v; size_t n, cap; } u64vec;
typedef struct { uint32_t *v; size_t n, cap; } u32vec;
typedef struct { uint8_t *v; size_t n, cap; } u8vec;
#define VPUSH(vec, T, x) do { if ((vec).n == (vec).cap) { (vec).cap = (vec).cap ? (vec).cap * 2 : 1024; \
    (vec).v = (T *)realloc((vec).v, (vec).cap * sizeof(T)); } (vec).v[(vec).n++] = (x); } while (0)

static int cmp_u64(const void *a, const void *b)
{
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : x > y;
}

typedef struct { uint64_t kmer; uint32_t idx; } kmer_idx;
static int cmp_kmer_idx(const void *a, const void *b)
{
    const kmer_idx *x = (const kmer_idx *)a, *y = (const kmer_idx *)b;
    if (x->kmer != y->kmer) return x->kmer < y->kmer ? -1 : 1;
    return x->idx < y->idx ? -1 : x->idx > y->idx;
}

typedef struct {
    u64vec kmer; u32vec gpos; u32vec ctg; u8vec fwd; u64vec markers;
} sketch_acc;

/* FracMinHash over one kept record.  pos is the index of the k-mer's LAST base; the first
 * marker_k-1 positions only warm the rolling registers up (seed and marker k-mers are both
 * examined from i = marker_k-1 on), following skani's seeding loop structure.
 * A k-mer is sampled when (hash ^ window) < 2^64 / c as unsigned numbers: window = 0 keeps the
 * hashes [0, T); window = 2^63 keeps [2^63, 2^63 + T), which is what a SIGNED 64-bit compare of the
 * hash against i64::MIN + T keeps (p->sample_window; oracle/sample_hypotheses.py). */
static void sketch_range(const uint8_t *s, uint32_t from, uint32_t to, uint32_t first_pos, uint32_t ctg,
                         uint32_t goff, const oracle_params_t *p, sketch_acc *acc)
{
    const int k = p->k, mk = p->marker_k;
    const uint64_t smask = (k == 32) ? ~0ULL : ((1ULL << (2 * k)) - 1);
    const uint64_t mmask = (mk == 32) ? ~0ULL : ((1ULL << (2 * mk)) - 1);
    const int sshift = 2 * (k - 1), mshift = 2 * (mk - 1);
    const uint64_t sthr = UINT64_MAX / (uint64_t)p->c;
    const uint64_t mthr = UINT64_MAX / (uint64_t)p->marker_c;
    const uint64_t win = p->sample_window ? 0x8000000000000000ULL : 0;
    const int fst = p->hash_first_step;
    uint64_t fs = 0, rs = 0, fm = 0, rm = 0;
    const int enc = p->rule[7] & 1, cmax = (p->rule[7] >> 1) & 1;   /* hypotheses: A,C,T,G = 0..3; canonical = larger */
    for (uint32_t i = from; i < to; i++) {
        uint64_t nf = base_code(s[i]), nr = 3 - nf;
        if (enc) { nf = (nf == 2) ? 3 : (nf == 3) ? 2 : nf; nr = nf ^ 2; }
        fs = ((fs << 2) | nf) & smask;
        rs = (rs >> 2) | (nr << sshift);
        fm = ((fm << 2) | nf) & mmask;
        rm = (rm >> 2) | (nr << mshift);
        if (i < first_pos) continue;
        int fwd = cmax ? fs > rs : fs < rs;
        uint64_t cs = fwd ? fs : rs;
        if ((mm_hash64_v(cs, fst) ^ win) < sthr) {
            VPUSH(acc->kmer, uint64_t, cs);
            VPUSH(acc->gpos, uint32_t, goff + i);
            VPUSH(acc->ctg, uint32_t, ctg);
            VPUSH(acc->fwd, uint8_t, (uint8_t)fwd);
        }
        uint64_t cm = (cmax ? fm > rm : fm < rm) ? fm : rm;
        if ((mm_hash64_v(cm, fst) ^ win) < mthr) VPUSH(acc->markers, uint64_t, cm);
    }
}

static void sketch_contig(const uint8_t *s, uint32_t len, uint32_t ctg, uint32_t goff,
                          const oracle_params_t *p, sketch_acc *acc)
{
    const uint32_t mk = (uint32_t)p->marker_k;
    if (len < mk) return;
    if (p->quarters == 1) {
        /* hypothesis: four SIMD lanes take the k-mers of a record in four equal runs; the last
         * (number of k-mers) mod 4 k-mers are not examined */
        uint32_t q = (len - mk + 1) / 4;
        for (uint32_t j = 0; j < 4; j++)
            sketch_range(s, j * q, (j + 1) * q + mk - 1, j * q + mk - 1, ctg, goff, p, acc);
    } else if (p->quarters == 2) {
        /* hypothesis: four disjoint quarters of len / 4 bases, each warmed up on its own */
        uint32_t q = len / 4;
        if (q < mk) return;
        for (uint32_t j = 0; j < 4; j++)
            sketch_range(s, j * q, (j + 1) * q, j * q + mk - 1, ctg, goff, p, acc);
    } else {
        sketch_range(s, 0, len, mk - 1, ctg, goff, p, acc);
    }
}

static void genome_finish(oracle_genome_t *g, sketch_acc *acc, const oracle_params_t *p)
{
    g->n_seeds = (uint32_t)acc->kmer.n;
    g->s_kmer = acc->kmer.v; g->s_gpos = acc->gpos.v; g->s_ctg = acc->ctg.v; g->s_fwd = acc->fwd.v;
    g->by_kmer = (uint32_t *)malloc((g->n_seeds + 1) * sizeof(uint32_t));
    kmer_idx *ki = (kmer_idx *)malloc((g->n_seeds + 1) * sizeof(kmer_idx));
    for (uint32_t i = 0; i < g->n_seeds; i++) { ki[i].kmer = g->s_kmer[i]; ki[i].idx = i; }
    qsort(ki, g->n_seeds, sizeof(kmer_idx), cmp_kmer_idx);
    for (uint32_t i = 0; i < g->n_seeds; i++) g->by_kmer[i] = ki[i].idx;
    free(ki);
    /* repetitive k-mer cut-off: with D distinct seed k-mers, the multiplicity of
     * rank D - D/1000 - 1 (ascending); the filter is off when that is below rep_floor. */
    {
        u64vec cnt = {0};
        for (uint32_t i = 0; i < g->n_seeds;) {
            uint32_t j = i;
            while (j < g->n_seeds && g->s_kmer[g->by_kmer[j]] == g->s_kmer[g->by_kmer[i]]) j++;
            VPUSH(cnt, uint64_t, (uint64_t)(j - i < ANI_REP_HIST - 1 ? j - i : ANI_REP_HIST - 1));
            i = j;
        }
        g->rep_cut = UINT32_MAX;
        if (cnt.n) {
            qsort(cnt.v, cnt.n, sizeof(uint64_t), cmp_u64);
            uint64_t cut = cnt.v[cnt.n - cnt.n / 1000 - 1];
            if (cut >= (uint64_t)p->rep_floor) g->rep_cut = (uint32_t)cut;
        }
        free(cnt.v);
    }
    qsort(acc->markers.v, acc->markers.n, sizeof(uint64_t), cmp_u64);
    size_t m = 0;
    for (size_t i = 0; i < acc->markers.n; i++)
        if (m == 0 || acc->markers.v[i] != acc->markers.v[m - 1]) acc->markers.v[m++] = acc->markers.v[i];
    g->markers = acc->markers.v;
    g->n_markers = (uint32_t)m;
}

/* N50 exactly as /root/reference/src/skDER/util.py:686-724: all records, half = int(sum/2),
 * lengths descending, first cumulative sum >= half. */
static uint64_t n50_of(uint64_t *lens, size_t n)
{
    if (!n) return 0;
    qsort(lens, n, sizeof(uint64_t), cmp_u64);
    uint64_t tot = 0;
    for (size_t i = 0; i < n; i++) tot += lens[i];
    uint64_t half = tot / 2, cum = 0;
    for (size_t i = n; i-- > 0;) {
        cum += lens[i];
        if (cum >= half) return lens[i];
    }
    return lens[0];
}

oracle_genome_t *oracle_genome_from_bases(const uint8_t *bases, const uint32_t *lens, uint32_t n_records,
                                          const char *file_name, const char *first_name,
                                          const oracle_params_t *p)
{
    oracle_genome_t *g = (oracle_genome_t *)calloc(1, sizeof(*g));
    sketch_acc acc;
    memset(&acc, 0, sizeof(acc));
    g->file_name = strdup(file_name ? file_name : "");
    g->first_name = strdup(first_name ? first_name : "");
    g->ctg_off = (uint32_t *)malloc((n_records + 1) * sizeof(uint32_t));
    uint64_t *all = (uint64_t *)malloc((n_records + 1) * sizeof(uint64_t));
    uint64_t off = 0;
    uint32_t goff = 0;
    for (uint32_t r = 0; r < n_records; r++) {
        all[r] = lens[r];
        if (lens[r] >= (uint32_t)p->min_contig) {
            g->ctg_off[g->n_contigs] = goff;
            sketch_contig(bases + off, lens[r], g->n_contigs, goff, p, &acc);
            g->n_contigs++;
            goff += lens[r];
        }
        off += lens[r];
    }
    g->ctg_off[g->n_contigs] = goff;
    g->total_len = goff;
    g->n50 = n50_of(all, n_records);
    free(all);
    genome_finish(g, &acc, p);
    return g;
}

oracle_genome_t *oracle_genome_load(const char *path, const oracle_params_t *p, char *err, size_t errlen)
{
    gzFile f = gzopen(path, "rb");
    if (!f) {
        if (err) snprintf(err, errlen, "cannot open %s", path);
        return NULL;
    }
    gzbuffer(f, 1 << 20);
    u8vec seq = {0};
    u32vec lens = {0};
    char *first_name = NULL;      /* first record >= min_contig */
    char *cur_name = NULL;
    size_t cur_len = 0;
    int have_rec = 0;
    size_t cap = 1 << 16;
    char *line = (char *)malloc(cap);
    for (;;) {
        size_t n = 0;
        int eof = 0;
        for (;;) { /* one full line of arbitrary length */
            if (!gzgets(f, line + n, (int)(cap - n))) { eof = (n == 0); break; }
            n += strlen(line + n);
            if (n && line[n - 1] == '\n') break;
            if (n + 1 >= cap) { cap *= 2; line = (char *)realloc(line, cap); }
            else if (gzeof(f)) break;
        }
        if (eof) break;
        while (n && (line[n - 1] == '\n' || line[n - 1] == '\r')) line[--n] = 0;
        if (line[0] == '>') {
            if (have_rec) {
                VPUSH(lens, uint32_t, (uint32_t)cur_len);
                if (!first_name && cur_len >= (size_t)p->min_contig) { first_name = cur_name; cur_name = NULL; }
            }
            free(cur_name);
            cur_name = strdup(line + 1);
            cur_len = 0;
            have_rec = 1;
        } else if (have_rec) {
            for (size_t i = 0; i < n; i++) {
                if (line[i] == ' ' || line[i] == '\t') continue;
                VPUSH(seq, uint8_t, (uint8_t)line[i]);
                cur_len++;
            }
        }
    }
    if (have_rec) {
        VPUSH(lens, uint32_t, (uint32_t)cur_len);
        if (!first_name && cur_len >= (size_t)p->min_contig) { first_name = cur_name; cur_name = NULL; }
    }
    free(cur_name);
    free(line);
    gzclose(f);
    if (!lens.n) {
        if (err) snprintf(err, errlen, "no FASTA records in %s", path);
        free(seq.v); free(lens.v); free(first_name);
        return NULL;
    }
    oracle_genome_t *g = oracle_genome_from_bases(seq.v, lens.v, (uint32_t)lens.n, path,
                                                  first_name ? first_name : "", p);
    free(seq.v); free(lens.v); free(first_name);
    return g;
}

void oracle_genome_free(oracle_genome_t *g)
{
    if (!g) return;
    free(g->file_name); free(g->first_name); free(g->ctg_off);
    free(g->s_kmer); free(g->s_gpos); free(g->s_ctg); free(g->s_fwd); free(g->by_kmer);
    free(g->markers);
    free(g);
}

uint32_t oracle_genome_n_seeds(const oracle_genome_t *g) { return g->n_seeds; }
uint32_t oracle_genome_n_markers(const oracle_genome_t *g) { return g->n_markers; }
uint32_t oracle_genome_n_contigs(const oracle_genome_t *g) { return g->n_contigs; }
uint64_t oracle_genome_total_len(const oracle_genome_t *g) { return g->total_len; }
uint64_t oracle_genome_n50(const oracle_genome_t *g) { return g->n50; }
uint32_t oracle_genome_rep_cut(const oracle_genome_t *g) { return g->rep_cut; }
const char *oracle_genome_name(const oracle_genome_t *g) { return g->first_name; }
void oracle_genome_seeds(const oracle_genome_t *g, uint64_t *kmer, uint32_t *gpos, uint32_t *contig, uint8_t *fwd)
{
    if (kmer) memcpy(kmer, g->s_kmer, g->n_seeds * sizeof(uint64_t));
    if (gpos) memcpy(gpos, g->s_gpos, g->n_seeds * sizeof(uint32_t));
    if (contig) memcpy(contig, g->s_ctg, g->n_seeds * sizeof(uint32_t));
    if (fwd) memcpy(fwd, g->s_fwd, g->n_seeds);
}
void oracle_genome_markers(const oracle_genome_t *g, uint64_t *markers)
{
    memcpy(markers, g->markers, g->n_markers * sizeof(uint64_t));
}
void oracle_genome_contig_offsets(const oracle_genome_t *g, uint32_t *off)
{
    memcpy(off, g->ctg_off, (g->n_contigs + 1) * sizeof(uint32_t));
}

/* ------------------------------------------------------------------ screen */

/* x^n by repeated multiplication in a fixed order (bit-reproducible on the device). */
static double powi_fixed(double x, int n)
{
    double r = 1.0;
    for (int i = 0; i < n; i++) r = r * x;
    return r;
}

/* marker containment screen: pass iff shared / min(|Ma|,|Mb|) > (s/100)^marker_k,
 * evaluated as shared > cutoff * min; marker sets below small_pass always pass. */
int oracle_screen(const oracle_genome_t *a, const oracle_genome_t *b, double screen_ani_pct,
                  const oracle_params_t *p, uint32_t *shared_out)
{
    uint32_t i = 0, j = 0, shared = 0;
    while (i < a->n_markers && j < b->n_markers) {
        if (a->markers[i] < b->markers[j]) i++;
        else if (a->markers[i] > b->markers[j]) j++;
        else { shared++; i++; j++; }
    }
    if (shared_out) *shared_out = shared;
    uint32_t mn = a->n_markers < b->n_markers ? a->n_markers : b->n_markers;
    if (screen_ani_pct <= 0.0) return 1;
    if (mn < (uint32_t)p->small_pass) return 1;
    double cutoff = powi_fixed(screen_ani_pct / 100.0, p->marker_k);
    return (double)shared > cutoff * (double)mn;
}

/* ------------------------------------------------------------------ k-th root, ANI model */

/* (num/den)^(1/k) by Newton iterations on doubles using only + - * / in a fixed order (compile
 * with -ffp-contract=off): bit-identical on host and device.  0 for an empty ratio, 1 when
 * num >= den. */
double oracle_root(uint64_t num, uint64_t den, int k)
{
    if (den == 0 || num == 0) return 0.0;
    if (num >= den) return 1.0;
    double x = (double)num / (double)den;
    double y = 1.0;
    double km1 = (double)(k - 1), kk = (double)k;
    for (int it = 0; it < ANI_ROOT_ITERS; it++) {
        double yp = 1.0;
        for (int i = 0; i < k - 1; i++) yp = yp * y;      /* y^(k-1) */
        y = (km1 * y + x / yp) / kk;
    }
    return y;
}

/* skani's default output passes its chunk-level k-mer ANI through a gradient-boosted regression
 * ("learned ANI") whose model is not reproducible here.  Stand-in: the two-parameter
 * line of include/skder_amd_spec.h in the divergences of the cell and the span estimate. */
double oracle_model_ani(double ani_cell, double ani_span)
{
    double d = ANI_CAL_CELL * (100.0 * (1.0 - ani_cell)) + ANI_CAL_SPAN * (100.0 * (1.0 - ani_span));
    double a = 1.0 - d / 100.0;
    if (a < 0.0) a = 0.0;
    if (a > 1.0) a = 1.0;
    return a;
}

/* ------------------------------------------------------------------ pair */

typedef struct {
    uint32_t qi;      /* index of the seed on the chunked genome (position order) */
    uint32_t qpos;    /* gpos on the chunked genome */
    uint32_t rpos;    /* gpos on the other genome */
    uint32_t rctg;
    uint8_t rev;
} anchor_t;

/* Which genome is cut into chunks?  The one with the smaller T * (T / n_contigs) (shorter and
 * more fragmented); ties: fewer seeds, then fewer markers, then the `query` argument.
 * Returns 1 to chunk `query`. */
static int chunk_query(const oracle_genome_t *ref, const oracle_genome_t *query, const oracle_params_t *p)
{
    switch (p->rule[0]) {
    case 1: if (query->total_len != ref->total_len) return query->total_len < ref->total_len; break;
    case 2: if (query->n_seeds != ref->n_seeds) return query->n_seeds < ref->n_seeds; break;
    case 3: if (query->total_len != ref->total_len) return query->total_len > ref->total_len; break;
    case 4: return 1;
    case 5: return 0;
    default: break;
    }
    double tq = (double)query->total_len, tr = (double)ref->total_len;
    double sq = tq * (tq / (double)(query->n_contigs ? query->n_contigs : 1));
    double sr = tr * (tr / (double)(ref->n_contigs ? ref->n_contigs : 1));
    if (sq != sr) return sq < sr;
    /* equal scores: decide by content, not by argument order, so that swapping Ref and Query only
     * swaps the two AF columns */
    if (query->n_seeds != ref->n_seeds) return query->n_seeds < ref->n_seeds;
    if (query->n_markers != ref->n_markers) return query->n_markers < ref->n_markers;
    return 1;
}

static uint32_t lower_bound_kmer(const oracle_genome_t *g, uint64_t kmer)
{
    uint32_t lo = 0, hi = g->n_seeds;
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if (g->s_kmer[g->by_kmer[mid]] < kmer) lo = mid + 1; else hi = mid;
    }
    return lo;
}
static uint32_t multiplicity(const oracle_genome_t *g, uint64_t kmer, uint32_t *lo_out)
{
    uint32_t lo = lower_bound_kmer(g, kmer), hi = lo;
    while (hi < g->n_seeds && g->s_kmer[g->by_kmer[hi]] == kmer) hi++;
    if (lo_out) *lo_out = lo;
    return hi - lo;
}

typedef struct { int32_t score; uint32_t idx; } sc_idx;
static int cmp_sc_desc(const void *a, const void *b)
{
    const sc_idx *x = (const sc_idx *)a, *y = (const sc_idx *)b;
    if (x->score != y->score) return x->score > y->score ? -1 : 1;
    return x->idx < y->idx ? -1 : x->idx > y->idx; /* ties: earlier anchor first */
}
static int cmp_chain(const void *a, const void *b)
{
    const oracle_chain_t *x = (const oracle_chain_t *)a, *y = (const oracle_chain_t *)b;
    if (x->score != y->score) return x->score > y->score ? -1 : 1;
    if (x->q0 != y->q0) return x->q0 < y->q0 ? -1 : 1;
    if (x->r0 != y->r0) return x->r0 < y->r0 ? -1 : 1;
    if (x->q1 != y->q1) return x->q1 < y->q1 ? -1 : 1;
    return 0;
}

int oracle_pair(const oracle_genome_t *ref, const oracle_genome_t *query, const oracle_params_t *p,
                oracle_pair_t *out, oracle_chain_t *chains_out, uint32_t chain_cap)
{
    memset(out, 0, sizeof(*out));
    int cq = chunk_query(ref, query, p);
    const oracle_genome_t *Q = cq ? query : ref;   /* chunked genome */
    const oracle_genome_t *R = cq ? ref : query;
    out->chunked_query = cq;

    /* 1. anchors in chunked-genome position order; several hits of one seed in ascending gpos
     *    of the other genome.  k-mers more frequent than either genome's rep_cut are skipped. */
    size_t na = 0, cap = 1 << 15;
    anchor_t *A = (anchor_t *)malloc(cap * sizeof(anchor_t));
    for (uint32_t qi = 0; qi < Q->n_seeds; qi++) {
        uint64_t km = Q->s_kmer[qi];
        uint32_t lo, cnt = multiplicity(R, km, &lo);
        if (!cnt || cnt > R->rep_cut) continue;
        if (Q->rep_cut != UINT32_MAX && multiplicity(Q, km, NULL) > Q->rep_cut) continue;
        for (uint32_t t = lo; t < lo + cnt; t++) {
            uint32_t ri = R->by_kmer[t];
            if (na == cap) { cap *= 2; A = (anchor_t *)realloc(A, cap * sizeof(anchor_t)); }
            A[na].qi = qi;
            A[na].qpos = Q->s_gpos[qi];
            A[na].rpos = R->s_gpos[ri];
            A[na].rctg = R->s_ctg[ri];
            A[na].rev = (uint8_t)(Q->s_fwd[qi] != R->s_fwd[ri]);
            na++;
        }
    }
    out->n_anchors = (uint32_t)na;

    int32_t *f = (int32_t *)malloc((na + 1) * sizeof(int32_t));
    int32_t *bp = (int32_t *)malloc((na + 1) * sizeof(int32_t));
    uint8_t *used = (uint8_t *)calloc(na + 1, 1);
    sc_idx *order = (sc_idx *)malloc((na + 1) * sizeof(sc_idx));
    size_t nch = 0, chcap = 1024;
    oracle_chain_t *C = (oracle_chain_t *)malloc(chcap * sizeof(oracle_chain_t));
    uint32_t n_chunks = 0;
    size_t ckcap = 1024;
    uint32_t *cell_seeds = (uint32_t *)malloc(ckcap * sizeof(uint32_t));   /* per chunk with anchors: ALL seeds of its 20 kb cell */

    /* 2. chunks: (record, (gpos - record_off) / chunk_len) of the chunked genome */
    size_t s = 0;
    while (s < na) {
        uint32_t qc = Q->s_ctg[A[s].qi];
        uint32_t ck = (A[s].qpos - Q->ctg_off[qc]) / (uint32_t)p->chunk_len;
        size_t e = s;
        if (p->rule[1] == 1) {          /* hypothesis: a chunk begins at an anchor and holds the next chunk_len bases */
            while (e < na && Q->s_ctg[A[e].qi] == qc && A[e].qpos - A[s].qpos <= (uint32_t)p->chunk_len) e++;
        } else
        while (e < na && Q->s_ctg[A[e].qi] == qc &&
               (A[e].qpos - Q->ctg_off[qc]) / (uint32_t)p->chunk_len == ck) e++;
        /* 3. banded chaining inside [s, e): f[i] = max(score, max_j f[j] + score - |dq - dr|),
         *    nearest predecessor first, strict '>' keeps the nearest of equal candidates */
        for (size_t i = s; i < e; i++) {
            int32_t best = p->anchor_score;
            int32_t bj = -1;
            size_t jlo = (i - s > (size_t)p->band) ? i - (size_t)p->band : s;
            for (size_t j = i; j-- > jlo;) {
                int64_t dq = (int64_t)A[i].qpos - (int64_t)A[j].qpos;
                if (dq > p->bp_band) break;
                if (A[j].rctg != A[i].rctg || A[j].rev != A[i].rev) continue;
                int64_t dr = A[i].rev ? (int64_t)A[j].rpos - (int64_t)A[i].rpos
                                      : (int64_t)A[i].rpos - (int64_t)A[j].rpos;
                if (dq <= 0 || dr <= 0) continue;
                if (dq > p->max_lin || dr > p->max_lin) continue;
                int64_t gap = dq > dr ? dq - dr : dr - dq;
                if (gap > p->max_gap) continue;
                int32_t sc = f[j] + p->anchor_score - (int32_t)gap;
                if (sc > best) { best = sc; bj = (int32_t)j; }
            }
            f[i] = best;
            bp[i] = bj;
        }
        /* 4. chains of the chunk: ends by (score desc, index asc); back-track until the chain
         *    start or an anchor that an earlier chain took; fewer than min_anchors anchors ->
         *    not a chain, its anchors stay free */
        size_t m = e - s;
        for (size_t i = 0; i < m; i++) { order[i].score = f[s + i]; order[i].idx = (uint32_t)(s + i); }
        qsort(order, m, sizeof(sc_idx), cmp_sc_desc);
        for (size_t t = 0; t < m; t++) {
            uint32_t endi = order[t].idx;
            if (used[endi]) continue;
            uint32_t n = 0, first = endi, rmin = UINT32_MAX, rmax = 0;
            int32_t cur = (int32_t)endi;
            while (cur >= 0 && !used[cur]) {
                n++;
                first = (uint32_t)cur;
                if (A[cur].rpos < rmin) rmin = A[cur].rpos;
                if (A[cur].rpos > rmax) rmax = A[cur].rpos;
                cur = bp[cur];
            }
            if (p->rule[3] && cur >= 0) continue;      /* hypothesis: a chain that runs into a taken anchor is no chain */
            if (n < (uint32_t)p->min_anchors) {
                if (p->rule[2]) {                      /* hypothesis: its anchors are spent all the same */
                    cur = (int32_t)endi;
                    while (cur >= 0 && !used[cur]) { int32_t nx = bp[cur]; used[cur] = 1; cur = nx; }
                }
                continue;
            }
            if (p->rule[6] && t > 0 && nch > 0 && C[nch - 1].chunk == n_chunks) break;  /* hypothesis: one chain per chunk */
            cur = (int32_t)endi;
            while (cur >= 0 && !used[cur]) { int32_t nx = bp[cur]; used[cur] = 1; cur = nx; }
            if (nch == chcap) { chcap *= 2; C = (oracle_chain_t *)realloc(C, chcap * sizeof(oracle_chain_t)); }
            C[nch].score = order[t].score;
            C[nch].n_anchors = n;
            C[nch].n_seeds = A[endi].qi - A[first].qi + 1;
            C[nch].q0 = A[first].qpos; C[nch].q1 = A[endi].qpos;
            C[nch].r0 = rmin; C[nch].r1 = rmax;
            C[nch].rctg = A[endi].rctg;
            C[nch].kept = 1;
            C[nch].chunk = n_chunks;
            nch++;
        }
        /* seeds of the chunked genome in this cell: gpos in [cell start, cell end) of record qc */
        {
            uint32_t g0 = Q->ctg_off[qc] + ck * (uint32_t)p->chunk_len;
            uint64_t g1 = (uint64_t)g0 + (uint64_t)p->chunk_len;
            if (g1 > Q->ctg_off[qc + 1]) g1 = Q->ctg_off[qc + 1];
            uint32_t lo = A[s].qi, hi = A[e - 1].qi + 1;
            while (lo > 0 && Q->s_ctg[lo - 1] == qc && Q->s_gpos[lo - 1] >= g0) lo--;
            while (hi < Q->n_seeds && Q->s_ctg[hi] == qc && (uint64_t)Q->s_gpos[hi] < g1) hi++;
            if (n_chunks == ckcap) { ckcap *= 2; cell_seeds = (uint32_t *)realloc(cell_seeds, ckcap * sizeof(uint32_t)); }
            cell_seeds[n_chunks] = hi - lo;
        }
        n_chunks++;
        s = e;
    }
    out->n_chunks = n_chunks;
    out->n_chains_all = (uint32_t)nch;

    /* 5. across chunks: chains by (score desc, q0, r0); a chain is dropped when more than half
     *    of its span on the other genome is covered by ONE better kept chain on the same record */
    qsort(C, nch, sizeof(oracle_chain_t), cmp_chain);
    for (size_t i = 0; i < nch; i++) {
        uint32_t li = C[i].r1 - C[i].r0;
        for (size_t j = 0; j < i; j++) {
            if (!C[j].kept || C[j].rctg != C[i].rctg) continue;
            uint32_t lo = C[i].r0 > C[j].r0 ? C[i].r0 : C[j].r0;
            uint32_t hi = C[i].r1 < C[j].r1 ? C[i].r1 : C[j].r1;
            if (p->rule[4] == 1) break;                /* hypothesis: no overlap filter */
            if (p->rule[4] == 3 ? hi > lo :
                hi > lo && (uint64_t)ANI_REF_OVERLAP_DEN * (hi - lo) > (uint64_t)ANI_REF_OVERLAP_NUM * li) {
                C[i].kept = 0;
                break;
            }
        }
    }

    /* 6. sums over the kept chains; N = all seeds of the cells that hold a kept chain; the two
     *    k-mer estimates and the model (include/skder_amd_spec.h) */
    {
        uint8_t *cell_used = (uint8_t *)calloc(n_chunks + 1, 1);
        for (size_t i = 0; i < nch; i++) {
            if (!C[i].kept) continue;
            out->sum_seeds += C[i].n_seeds;
            out->sum_anchors += C[i].n_anchors;
            out->sum_span += p->rule[5] ? (uint64_t)(C[i].r1 - C[i].r0) : (uint64_t)(C[i].q1 - C[i].q0);
            out->n_chains++;
            if (!cell_used[C[i].chunk]) { cell_used[C[i].chunk] = 1; out->cell_seeds += cell_seeds[C[i].chunk]; }
        }
        free(cell_used);
    }
    out->aligned_bases = out->sum_span + (uint64_t)p->pad * out->n_chains;
    if (out->sum_seeds) {
        out->ani_raw = oracle_root(out->sum_anchors, out->cell_seeds, p->k);
        out->ani_span = oracle_root(out->sum_anchors, out->sum_seeds, p->k);
        out->ani = p->learned ? oracle_model_ani(out->ani_raw, out->ani_span) : out->ani_raw;
    }
    double B = (double)out->aligned_bases;
    double afq = Q->total_len ? B / (double)Q->total_len : 0.0;
    double afr = R->total_len ? B / (double)R->total_len : 0.0;
    if (afq > 1.0) afq = 1.0;
    if (afr > 1.0) afr = 1.0;
    if (cq) { out->af_query = afq; out->af_ref = afr; }
    else { out->af_query = afr; out->af_ref = afq; }

    if (chains_out)
        memcpy(chains_out, C, (nch < chain_cap ? nch : chain_cap) * sizeof(oracle_chain_t));
    free(A); free(f); free(bp); free(used); free(order); free(C); free(cell_seeds);
    return 0;
}

/* ------------------------------------------------------------------ drivers */

typedef struct {
    uint32_t i, j;      /* ref index, query index */
    float ani, af_ref, af_query;   /* fractions, stored single precision as skani's result struct does */
} edge_t;

/* hashbrown (SwissTable) + FxHash iteration-order model, SURVEY V2: bucket = (key*K) & mask,
 * tables grow 4 -> 8 -> 16 -> ... when items exceed 3, 7, 14, 28, ... ; iteration is ascending
 * bucket index; collisions resolved by the first free slot at/after the home slot (what
 * hashbrown's group probe yields while the first 16-wide group has a free slot). */
typedef struct { uint64_t *key; uint8_t *full; uint32_t buckets, items; } fxmap;
static uint32_t fx_capacity(uint32_t buckets) { return buckets < 8 ? buckets - 1 : buckets / 8 * 7; }
static void fx_place(fxmap *m, uint64_t key)
{
    uint64_t h = key * 0x517cc1b727220a95ULL;
    uint32_t mask = m->buckets - 1, pos = (uint32_t)(h & mask);
    while (m->full[pos]) pos = (pos + 1) & mask;
    m->full[pos] = 1; m->key[pos] = key;
}
static void fx_init(fxmap *m) { memset(m, 0, sizeof(*m)); }
static void fx_insert(fxmap *m, uint64_t key)
{
    if (m->buckets == 0 || m->items + 1 > fx_capacity(m->buckets)) {
        uint32_t nb = m->buckets ? m->buckets * 2 : 4;
        fxmap n; n.buckets = nb; n.items = m->items;
        n.key = (uint64_t *)calloc(nb, sizeof(uint64_t)); n.full = (uint8_t *)calloc(nb, 1);
        for (uint32_t b = 0; b < m->buckets; b++) if (m->full[b]) fx_place(&n, m->key[b]);
        free(m->key); free(m->full);
        *m = n;
    }
    fx_place(m, key); m->items++;
}
static void fx_free(fxmap *m) { free(m->key); free(m->full); }

static char **read_listing(const char *path, uint32_t *n_out, char *err, size_t errlen)
{
    FILE *f = fopen(path, "r");
    if (!f) { if (err) snprintf(err, errlen, "cannot open listing %s", path); return NULL; }
    size_t cap = 64, n = 0;
    char **v = (char **)malloc(cap * sizeof(char *));
    char *line = NULL; size_t lc = 0; ssize_t len;
    while ((len = getline(&line, &lc, f)) >= 0) {
        while (len && (line[len - 1] == '\n' || line[len - 1] == '\r' || line[len - 1] == ' ')) line[--len] = 0;
        if (!len) continue;
        if (n == cap) { cap *= 2; v = (char **)realloc(v, cap * sizeof(char *)); }
        v[n++] = strdup(line);
    }
    free(line); fclose(f);
    *n_out = (uint32_t)n;
    return v;
}

static int cmp_str(const void *a, const void *b) { return strcmp(*(char *const *)a, *(char *const *)b); }

static oracle_genome_t **load_all(char **paths, uint32_t n, int threads, const oracle_params_t *p,
                                  char *err, size_t errlen)
{
    oracle_genome_t **g = (oracle_genome_t **)calloc(n ? n : 1, sizeof(*g));
    int bad = 0;
    (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1)
#endif
    for (uint32_t i = 0; i < n; i++) {
        char e[256];
        g[i] = oracle_genome_load(paths[i], p, e, sizeof e);
        if (!g[i]) {
#ifdef _OPENMP
#pragma omp critical(oracle_err)
#endif
            { bad = 1; if (err) snprintf(err, errlen, "%s", e); }
        }
    }
    if (bad) {
        for (uint32_t i = 0; i < n; i++) oracle_genome_free(g[i]);
        free(g);
        return NULL;
    }
    return g;
}

static void print_row(FILE *o, const char *rf, const char *qf, const edge_t *e, const char *rn, const char *qn)
{
    fprintf(o, "%s\t%s\t%.2f\t%.2f\t%.2f\t%s\t%s\n", rf, qf, (double)(e->ani * 100.0f),
            (double)(e->af_ref * 100.0f), (double)(e->af_query * 100.0f), rn, qn);
}
static const char *HEADER = "Ref_file\tQuery_file\tANI\tAlign_fraction_ref\tAlign_fraction_query\tRef_name\tQuery_name\n";

/* write to out.tmp then rename: the caller's only success test is "output file exists"
 * (/root/reference/src/skDER/util.py:642-645), so no partial file may be left behind. */
static FILE *open_tmp(const char *out, char *tmp, size_t tl)
{
    snprintf(tmp, tl, "%s.tmp.%d", out, (int)getpid());
    return fopen(tmp, "w");
}

int oracle_triangle(const char *listing, double min_af_pct, double screen_pct, int threads,
                    const char *out_tsv, const oracle_params_t *p, char *err, size_t errlen)
{
    uint32_t n = 0;
    char **paths = read_listing(listing, &n, err, errlen);
    if (!paths) return 1;
    qsort(paths, n, sizeof(char *), cmp_str); /* genomes indexed by ascending path string (V2) */
    oracle_genome_t **g = load_all(paths, n, threads, p, err, errlen);
    if (!g) { for (uint32_t i = 0; i < n; i++) free(paths[i]); free(paths); return 2; }

    size_t ecap = 1024, ne = 0;
    edge_t *E = (edge_t *)malloc(ecap * sizeof(edge_t));
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1)
#endif
    for (uint32_t i = 0; i < n; i++) {
        for (uint32_t j = i + 1; j < n; j++) {
            if (!oracle_screen(g[i], g[j], screen_pct, p, NULL)) continue;
            oracle_pair_t r;
            oracle_pair(g[i], g[j], p, &r, NULL, 0);
            if (r.n_chains == 0 || !(r.ani > 0.0)) continue;
#ifdef _OPENMP
#pragma omp critical(oracle_edges)
#endif
            {
                if (ne == ecap) { ecap *= 2; E = (edge_t *)realloc(E, ecap * sizeof(edge_t)); }
                E[ne].i = i; E[ne].j = j;
                E[ne].ani = (float)r.ani; E[ne].af_ref = (float)r.af_ref; E[ne].af_query = (float)r.af_query;
                ne++;
            }
        }
    }
    /* row order: outer map keyed by i, inner maps keyed by j (inserted ascending), bucket order */
    uint32_t *first = (uint32_t *)calloc(n + 1, sizeof(uint32_t));
    edge_t *S = (edge_t *)malloc((ne ? ne : 1) * sizeof(edge_t));
    for (size_t t = 0; t < ne; t++) first[E[t].i + 1]++;
    for (uint32_t i = 0; i < n; i++) first[i + 1] += first[i];
    {
        uint32_t *fill = (uint32_t *)calloc(n + 1, sizeof(uint32_t));
        for (size_t t = 0; t < ne; t++) S[first[E[t].i] + fill[E[t].i]++] = E[t];
        free(fill);
    }
    char tmp[4096];
    FILE *o = open_tmp(out_tsv, tmp, sizeof tmp);
    if (!o) {
        if (err) snprintf(err, errlen, "cannot write %s", out_tsv);
        goto fail;
    }
    fputs(HEADER, o);
    {
        fxmap outer; fx_init(&outer);
        for (uint32_t i = 0; i < n; i++) if (first[i + 1] > first[i]) fx_insert(&outer, i);
        for (uint32_t b = 0; b < outer.buckets; b++) {
            if (!outer.full[b]) continue;
            uint32_t i = (uint32_t)outer.key[b];
            /* inner: insert ascending j */
            fxmap inner; fx_init(&inner);
            /* selection sort by j is O(m^2); m <= n. Use simple insertion into a j-indexed lookup. */
            uint32_t m = first[i + 1] - first[i];
            edge_t *row = S + first[i];
            /* sort row by j ascending */
            for (uint32_t a = 1; a < m; a++) {
                edge_t x = row[a]; uint32_t c = a;
                while (c > 0 && row[c - 1].j > x.j) { row[c] = row[c - 1]; c--; }
                row[c] = x;
            }
            for (uint32_t a = 0; a < m; a++) fx_insert(&inner, row[a].j);
            for (uint32_t bb = 0; bb < inner.buckets; bb++) {
                if (!inner.full[bb]) continue;
                uint32_t j = (uint32_t)inner.key[bb];
                /* binary search j in row */
                uint32_t lo = 0, hi = m;
                while (lo < hi) { uint32_t mid = (lo + hi) / 2; if (row[mid].j < j) lo = mid + 1; else hi = mid; }
                const edge_t *e = &row[lo];
                double mx = e->af_ref > e->af_query ? e->af_ref : e->af_query;
                if (mx * 100.0 < min_af_pct) continue; /* V4: max(AF) on unrounded values */
                print_row(o, paths[i], paths[j], e, g[i]->first_name, g[j]->first_name);
            }
            fx_free(&inner);
        }
        fx_free(&outer);
    }
    fclose(o);
    if (rename(tmp, out_tsv) != 0) {
        if (err) snprintf(err, errlen, "cannot rename to %s", out_tsv);
        remove(tmp);
        goto fail;
    }
    free(first); free(S); free(E);
    for (uint32_t i = 0; i < n; i++) { oracle_genome_free(g[i]); free(paths[i]); }
    free(g); free(paths);
    return 0;
fail:
    free(first); free(S); free(E);
    for (uint32_t i = 0; i < n; i++) { oracle_genome_free(g[i]); free(paths[i]); }
    free(g); free(paths);
    return 3;
}

static int cmp_edge_ani_desc(const void *a, const void *b)
{
    const edge_t *x = (const edge_t *)a, *y = (const edge_t *)b;
    if (x->ani != y->ani) return x->ani > y->ani ? -1 : 1;
    return x->i < y->i ? -1 : x->i > y->i;
}

/* rectangle: every query (in listing order) against every reference; per query, rows by ANI
 * descending (verified on G4). Used for `dist` and `search`. */
static int rectangle(char **rpaths, uint32_t nr, char **qpaths, uint32_t nq, double min_af_pct,
                     double screen_pct, int threads, const char *out_tsv, const oracle_params_t *p,
                     char *err, size_t errlen)
{
    oracle_genome_t **R = load_all(rpaths, nr, threads, p, err, errlen);
    if (!R) return 2;
    oracle_genome_t **Q = load_all(qpaths, nq, threads, p, err, errlen);
    if (!Q) { for (uint32_t i = 0; i < nr; i++) oracle_genome_free(R[i]); free(R); return 2; }
    char tmp[4096];
    FILE *o = open_tmp(out_tsv, tmp, sizeof tmp);
    int rc = 0;
    if (!o) { if (err) snprintf(err, errlen, "cannot write %s", out_tsv); rc = 3; }
    else {
        fputs(HEADER, o);
        edge_t *E = (edge_t *)malloc((nr ? nr : 1) * sizeof(edge_t));
        for (uint32_t q = 0; q < nq; q++) {
            uint32_t ne = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1)
#endif
            for (uint32_t r = 0; r < nr; r++) {
                if (!oracle_screen(R[r], Q[q], screen_pct, p, NULL)) continue;
                oracle_pair_t pr;
                oracle_pair(R[r], Q[q], p, &pr, NULL, 0);
                if (pr.n_chains == 0 || !(pr.ani > 0.0)) continue;
                double mx = pr.af_ref > pr.af_query ? pr.af_ref : pr.af_query;
                if (mx * 100.0 < min_af_pct) continue;
#ifdef _OPENMP
#pragma omp critical(oracle_edges)
#endif
                {
                    E[ne].i = r; E[ne].j = q; E[ne].ani = (float)pr.ani;
                    E[ne].af_ref = (float)pr.af_ref; E[ne].af_query = (float)pr.af_query; ne++;
                }
            }
            qsort(E, ne, sizeof(edge_t), cmp_edge_ani_desc);
            for (uint32_t t = 0; t < ne; t++)
                print_row(o, rpaths[E[t].i], qpaths[q], &E[t], R[E[t].i]->first_name, Q[q]->first_name);
        }
        free(E);
        fclose(o);
        if (rename(tmp, out_tsv) != 0) { if (err) snprintf(err, errlen, "cannot rename to %s", out_tsv); remove(tmp); rc = 3; }
    }
    for (uint32_t i = 0; i < nr; i++) oracle_genome_free(R[i]);
    for (uint32_t i = 0; i < nq; i++) oracle_genome_free(Q[i]);
    free(R); free(Q);
    return rc;
}

int oracle_dist(const char *ref_listing, const char *query_listing, double min_af_pct, double screen_pct,
                int threads, const char *out_tsv, const oracle_params_t *p, char *err, size_t errlen)
{
    uint32_t nr = 0, nq = 0;
    char **r = read_listing(ref_listing, &nr, err, errlen);
    if (!r) return 1;
    char **q = read_listing(query_listing, &nq, err, errlen);
    if (!q) { for (uint32_t i = 0; i < nr; i++) free(r[i]); free(r); return 1; }
    int rc = rectangle(r, nr, q, nq, min_af_pct, screen_pct, threads, out_tsv, p, err, errlen);
    for (uint32_t i = 0; i < nr; i++) free(r[i]);
    for (uint32_t i = 0; i < nq; i++) free(q[i]);
    free(r); free(q);
    return rc;
}

int oracle_search(const char *listing_db, const char *query_path, double min_af_pct, double screen_pct,
                  int threads, const char *out_tsv, const oracle_params_t *p, char *err, size_t errlen)
{
    uint32_t nr = 0;
    char **r = read_listing(listing_db, &nr, err, errlen);
    if (!r) return 1;
    char *q[1];
    q[0] = (char *)query_path;
    int rc = rectangle(r, nr, q, 1, min_af_pct, screen_pct, threads, out_tsv, p, err, errlen);
    for (uint32_t i = 0; i < nr; i++) free(r[i]);
    free(r);
    return rc;
}
