// api.hip -- the C ABI of include/skder_amd.h.
#include <algorithm>
#include <chrono>
#include <numeric>
#include <thread>

#include "device_utils.h"
#include "engine.h"
#include <atomic>
#include "host_io.h"
#include "screen.h"
#include "synth.h"

static void set_err(char *err, size_t errlen, const std::string &msg)
{
    if (err && errlen) snprintf(err, errlen, "%s", msg.c_str());
}

#define API_TRY try {
#define API_CATCH_CTX(ctx_, rc_)                                                   \
    }                                                                              \
    catch (const std::exception &e) { if (ctx_) (ctx_)->last_error = e.what(); return rc_; }

// ---------------------------------------------------------------------------------------------
// context

extern "C" skder_ctx_t *skder_amd_ctx_create(int device, char *err, size_t errlen)
{
    skder_ctx *ctx = nullptr;
    try {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n == 0)
            throw SkError("no HIP device available: libskder_amd has no CPU fallback");
        if (device < 0 || device >= n) throw SkError("device index out of range");
        HIPCHECK(hipSetDevice(device));
        hipDeviceProp_t prop;
        HIPCHECK(hipGetDeviceProperties(&prop, device));
        if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
            throw SkError(std::string("libskder_amd is built for gfx950 (MI355X) only; device is ") + prop.gcnArchName);
        ctx = new skder_ctx();
        ctx->device = device;
#ifdef SKDER_CU_MASK_PROBE
        // (measurement build, profiles/run/r5_cu_mask.sh: the context's main queue confined to the first SKDER_AMD_CU_MASK compute units of
        // every group of 256 mask bits -- how do the chain stage's kernels scale with the CUs they get?)
        if (const char *e = getenv("SKDER_AMD_CU_MASK")) {
            const int ncu = atoi(e);
            uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            const char *mode = getenv("SKDER_AMD_CU_MASK_MODE");
            for (int c = 0; c < 256; c++) {
                // mode "spread": every k-th bit; default: the first ncu bits
                const bool on = (mode && mode[0] == 's') ? ((long long)c * ncu / 256 != (long long)(c + 1) * ncu / 256) : c < ncu;
                if (on) mask[c >> 5] |= 1u << (c & 31);
            }
            HIPCHECK(hipExtStreamCreateWithCUMask(&ctx->stream, 8, mask));
        } else
#endif
        HIPCHECK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        HIPCHECK(hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
        for (auto &e : ctx->ev) HIPCHECK(hipEventCreate(&e));
        HIPCHECK(hipMalloc(&ctx->d_flags, 64));
        HIPCHECK(hipMemset(ctx->d_flags, 0, 64));
        return ctx;
    } catch (const std::exception &e) {
        set_err(err, errlen, e.what());
        delete ctx;
        return nullptr;
    }
}

extern "C" void skder_amd_ctx_destroy(skder_ctx_t *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->stream2) (void)hipStreamSynchronize(ctx->stream2);
    if (ctx->chain_work && ctx->chain_work_free) ctx->chain_work_free(ctx->chain_work);
    for (auto &e : ctx->ev) (void)hipEventDestroy(e);
    if (ctx->d_flags) (void)hipFree(ctx->d_flags);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
    delete ctx;
}

extern "C" void *skder_amd_ctx_stream(skder_ctx_t *ctx) { return ctx ? (void *)ctx->stream : nullptr; }
extern "C" const char *skder_amd_last_error(skder_ctx_t *ctx) { return ctx ? ctx->last_error.c_str() : "null context"; }
extern "C" int skder_amd_last_timing(skder_ctx_t *ctx, double *out8)
{
    if (!ctx || !out8) return 1;
    for (int i = 0; i < 8; i++) out8[i] = ctx->timing[i];
    return 0;
}
extern "C" int skder_amd_copy_d2d(skder_ctx_t *ctx, void *dst, const void *src, size_t bytes)
{
    if (!ctx || (bytes && (!dst || !src))) return 1;
    API_TRY
    HIPCHECK(hipSetDevice(ctx->device));
    if (bytes) HIPCHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    return 0;
    API_CATCH_CTX(ctx, 2)
}

// memory the library keeps between calls (the ingest's pinned staging buffers and their device copies, the device
// allocator's cached blocks, the row orders' spare host list): handed back.  Buffers in use by a running call stay.
extern "C" int skder_amd_release_cached_buffers(int device)
{
    try {
        HIPCHECK(hipSetDevice(device));
        const bool ok = staging_release(device);
        pool_trim();
        rows_order_release_spare();
        return ok ? 0 : 1;
    } catch (const std::exception &) { return 2; }
}

extern "C" double skder_amd_last_index_ms(skder_ctx_t *ctx) { return ctx ? ctx->timing_index : 0.0; }
extern "C" double skder_amd_last_runs_ms(skder_ctx_t *ctx) { return ctx ? ctx->timing_runs : 0.0; }

extern "C" int skder_amd_last_counters(skder_ctx_t *ctx, uint64_t *out4)
{
    if (!ctx || !out4) return 1;
    for (int i = 0; i < 4; i++) out4[i] = ctx->counters[i];
    out4[2] = (uint64_t)(ctx->timing_join * 1000.0);   // join kernel time, microseconds
    return 0;
}

// ---------------------------------------------------------------------------------------------
// sketch sets

extern "C" skder_sketches_t *skder_amd_sketches_new(skder_ctx_t *ctx)
{
    if (!ctx) return nullptr;
    skder_sketches *s = new skder_sketches();
    s->ctx = ctx;
    return s;
}
extern "C" void skder_amd_sketches_free(skder_sketches_t *s)
{
    if (!s) return;
    (void)hipSetDevice(s->ctx->device);
    (void)hipStreamSynchronize(s->ctx->stream);
    // an index build left pending (skder_amd_sketches_index_part, or a triangle / rectangle call that failed half way)
    // still writes this set's buffers from the second queue: they go back to the allocator only once it is done
    if (s->ctx->stream2) (void)hipStreamSynchronize(s->ctx->stream2);
    if (s->idx_stream && s->idx_stream != s->ctx->stream && s->idx_stream != s->ctx->stream2) (void)hipStreamSynchronize(s->idx_stream);
    delete s;
}

extern "C" int skder_amd_sketch_batch(skder_sketches_t *s, const uint8_t *d_bases, const skder_batch_t *batch)
{
    if (!s || !batch) return 1;
    API_TRY
    HIPCHECK(hipSetDevice(s->ctx->device));
    sketch_batch_impl(s, d_bases, batch);
    return 0;
    API_CATCH_CTX(s->ctx, 2)
}

extern "C" int skder_amd_sketches_reserve(skder_sketches_t *s, uint64_t n_seeds, uint64_t n_markers)
{
    if (!s) return 1;
    API_TRY
    HIPCHECK(hipSetDevice(s->ctx->device));
    hipStream_t st = s->ctx->stream;
    s->seed_kmer.reserve(n_seeds, s->seed_kmer.n, st);
    s->seed_gpos.reserve(n_seeds + 32, s->seed_gpos.n, st);
    s->seed_ctg.reserve(n_seeds, s->seed_ctg.n, st);
    s->markers.reserve(n_markers, s->markers.n, st);
    return 0;
    API_CATCH_CTX(s->ctx, 2)
}

extern "C" int skder_amd_sketches_view(skder_sketches_t *s, skder_raw_view_t *out)
{
    if (!s || !out) return 1;
    out->n_genomes = s->n_genomes;
    out->n_seeds = s->h_seed_off.back();
    out->n_markers = s->h_marker_off.back();
    out->n_rec_goff = s->h_rec_goff.size();
    out->d_seed_kmer = s->seed_kmer.p; out->d_seed_gpos = s->seed_gpos.p; out->d_seed_ctg = s->seed_ctg.p;
    out->d_markers = s->markers.p;
    out->h_seed_off = s->h_seed_off.data(); out->h_marker_off = s->h_marker_off.data();
    out->h_genome_len = s->h_genome_len.data(); out->h_genome_nrec = s->h_genome_nrec.data();
    out->h_rec_goff = s->h_rec_goff.data();
    return 0;
}

// record index of every appended seed from its genome-linear position (binary search in the genome's
// record offsets): lets the sketch exchange leave the third of its volume that is derivable at home
struct CtgJob { uint64_t seed_begin, seed_end; uint32_t rec_begin, n_rec; };
__global__ __launch_bounds__(256) void ctg_from_gpos_kernel(const CtgJob *__restrict__ jobs, const uint32_t *__restrict__ rec_goff,
                                                            const uint32_t *__restrict__ gpos, uint32_t *__restrict__ ctg)
{
    const CtgJob j = jobs[blockIdx.x];
    const uint32_t *rg = rec_goff + j.rec_begin;
    for (uint64_t s = j.seed_begin + threadIdx.x; s < j.seed_end; s += 256) {
        const uint32_t p = gpos[s];
        uint32_t lo = 0, hi = j.n_rec;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (rg[mid] <= p) lo = mid; else hi = mid;
        }
        ctg[s] = lo;
    }
}

extern "C" int skder_amd_sketches_append_raw(skder_sketches_t *s, const skder_raw_view_t *raw)
{
    if (!s || !raw) return 1;
    API_TRY
    if (s->indexed || s->index_pending) throw SkError("sketch set already indexed; cannot append");
    HIPCHECK(hipSetDevice(s->ctx->device));
    hipStream_t st = s->ctx->stream;
    const uint64_t sb = s->seed_kmer.n, mb = s->markers.n;
    s->seed_kmer.resize(sb + raw->n_seeds, st);
    s->seed_gpos.reserve(sb + raw->n_seeds + 32, s->seed_gpos.n, st);
    s->seed_gpos.resize(sb + raw->n_seeds, st);
    s->seed_ctg.resize(sb + raw->n_seeds, st);
    s->markers.resize(mb + raw->n_markers, st);
    if (raw->n_seeds) {
        HIPCHECK(hipMemcpyAsync(s->seed_kmer.p + sb, raw->d_seed_kmer, raw->n_seeds * 4, hipMemcpyDefault /* the source may live on another GPU */, st));
        HIPCHECK(hipMemcpyAsync(s->seed_gpos.p + sb, raw->d_seed_gpos, raw->n_seeds * 4, hipMemcpyDefault /* the source may live on another GPU */, st));
        if (raw->d_seed_ctg)
            HIPCHECK(hipMemcpyAsync(s->seed_ctg.p + sb, raw->d_seed_ctg, raw->n_seeds * 4, hipMemcpyDefault /* the source may live on another GPU */, st));
    }
    if (raw->n_seeds && !raw->d_seed_ctg) {
        // no record indices given: derive them from the positions and the record tables
        std::vector<CtgJob> jobs(raw->n_genomes);
        std::vector<uint32_t> goff;
        size_t rgi = 0;
        for (uint32_t g = 0; g < raw->n_genomes; g++) {
            CtgJob &j = jobs[g];
            j.seed_begin = sb + raw->h_seed_off[g] - raw->h_seed_off[0];
            j.seed_end = sb + raw->h_seed_off[g + 1] - raw->h_seed_off[0];
            j.rec_begin = (uint32_t)goff.size(); j.n_rec = raw->h_genome_nrec[g];
            for (uint32_t r = 0; r <= raw->h_genome_nrec[g]; r++) goff.push_back(raw->h_rec_goff[rgi++]);
        }
        DevBuf<CtgJob> d_jobs;
        DevBuf<uint32_t> d_goff;
        d_jobs.resize(jobs.size(), st); d_goff.resize(goff.size() + 1, st);
        HIPCHECK(hipMemcpyAsync(d_jobs.p, jobs.data(), jobs.size() * sizeof(CtgJob), hipMemcpyHostToDevice, st));
        HIPCHECK(hipMemcpyAsync(d_goff.p, goff.data(), goff.size() * 4, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(ctg_from_gpos_kernel, dim3(raw->n_genomes), dim3(256), 0, st, d_jobs.p, d_goff.p, s->seed_gpos.p, s->seed_ctg.p);
        HIPCHECK(hipGetLastError());
        HIPCHECK(hipStreamSynchronize(st));
    }
    if (raw->n_markers)
        HIPCHECK(hipMemcpyAsync(s->markers.p + mb, raw->d_markers, raw->n_markers * 8, hipMemcpyDefault /* the source may live on another GPU */, st));
    HIPCHECK(hipStreamSynchronize(st));
    size_t rg = 0;
    for (uint32_t g = 0; g < raw->n_genomes; g++) {
        s->h_seed_off.push_back(sb + raw->h_seed_off[g + 1] - raw->h_seed_off[0]);
        s->h_marker_off.push_back(mb + raw->h_marker_off[g + 1] - raw->h_marker_off[0]);
        s->h_genome_len.push_back(raw->h_genome_len[g]);
        s->h_genome_nrec.push_back(raw->h_genome_nrec[g]);
        for (uint32_t r = 0; r <= raw->h_genome_nrec[g]; r++) s->h_rec_goff.push_back(raw->h_rec_goff[rg++]);
    }
    s->n_genomes += raw->n_genomes;
    return 0;
    API_CATCH_CTX(s->ctx, 2)
}

extern "C" int skder_amd_sketches_index(skder_sketches_t *s)
{
    if (!s) return 1;
    API_TRY
    HIPCHECK(hipSetDevice(s->ctx->device));
    index_impl(s);
    return 0;
    API_CATCH_CTX(s->ctx, 2)
}

static void settle_index(skder_sketches *s);     // an index build left pending by skder_amd_sketches_index_part: wait for it

// debugging / parity-test accessors: copy a genome's index-stage products to the host
extern "C" int skder_amd_debug_genome(skder_sketches_t *s, uint32_t g, uint32_t *n_chunks, uint32_t *rep_cut, uint32_t *bucket_bits,
                                      uint32_t *h_skmer, uint32_t *h_sgpos, uint32_t *h_sctg, uint32_t *h_pchunk)
{
    if (!s || g >= s->n_genomes) return 1;
    API_TRY
    settle_index(s);
    if (!s->indexed) return 1;
    HIPCHECK(hipSetDevice(s->ctx->device));
    const GenomeMeta &m = s->h_meta[g];
    if (n_chunks) *n_chunks = m.n_chunks;
    if (rep_cut) *rep_cut = m.rep_cut;
    if (bucket_bits) *bucket_bits = m.bucket_bits;
    if (h_skmer) HIPCHECK(hipMemcpy(h_skmer, s->skmer.p + m.seed_off, m.n_seeds * 4ull, hipMemcpyDeviceToHost));
    if (h_sgpos) HIPCHECK(hipMemcpy(h_sgpos, s->sgpos.p + m.seed_off, m.n_seeds * 4ull, hipMemcpyDeviceToHost));
    if (h_sctg) HIPCHECK(hipMemcpy(h_sctg, s->sctg.p + m.seed_off, m.n_seeds * 4ull, hipMemcpyDeviceToHost));
    if (h_pchunk) HIPCHECK(hipMemcpy(h_pchunk, s->pchunk.p + m.seed_off, m.n_seeds * 4ull, hipMemcpyDeviceToHost));
    return 0;
    API_CATCH_CTX(s->ctx, 2)
}

extern "C" int skder_amd_triangle_rows(skder_sketches_t *s, uint32_t row_begin, uint32_t row_stride, double screen_pct,
                                       const skder_edge_t **edges, uint64_t *n_edges)
{
    if (!s) return 1;
    API_TRY
    HIPCHECK(hipSetDevice(s->ctx->device));
    triangle_rows_impl(s, row_begin, row_stride, screen_pct);
    if (edges) *edges = s->ctx->edges.data();
    if (n_edges) *n_edges = s->ctx->edges.size();
    return 0;
    API_CATCH_CTX(s->ctx, 2)
}

extern "C" int skder_amd_sketches_index_part(skder_sketches_t *s, const uint8_t *full)
{
    if (!s) return 1;
    API_TRY
    HIPCHECK(hipSetDevice(s->ctx->device));
    if (s->indexed || s->index_pending) throw SkError("sketch set already indexed");
    // enqueued on the second queue and left pending: skder_amd_screen_rows, which does not read the index, runs beside it;
    // the calls that need the index or its per-genome results (rep_cuts, pairs_probed, chain_pairs, ...) wait for it
    index_begin(s, s->ctx->stream2 ? s->ctx->stream2 : s->ctx->stream, full);
    return 0;
    API_CATCH_CTX(s->ctx, 2)
}

static void settle_index(skder_sketches *s)
{
    if (s->index_pending) {
        HIPCHECK(hipSetDevice(s->ctx->device));
        index_finish(s);
    }
}

extern "C" int skder_amd_sketches_rep_cuts(skder_sketches_t *s, uint32_t *out)
{
    if (!s || !out) return 1;
    API_TRY
    settle_index(s);
    if (!s->indexed) return 1;
    for (uint32_t g = 0; g < s->n_genomes; g++) out[g] = s->full_index[g] ? s->h_meta[g].rep_cut : 0xFFFFFFFFu;
    return 0;
    API_CATCH_CTX(s->ctx, 2)
}

extern "C" int skder_amd_sketches_set_rep_cuts(skder_sketches_t *s, const uint32_t *in)
{
    if (!s || !in) return 1;
    API_TRY
    HIPCHECK(hipSetDevice(s->ctx->device));
    settle_index(s);
    std::vector<uint8_t> mask(s->n_genomes);
    for (uint32_t g = 0; g < s->n_genomes; g++) mask[g] = s->full_index[g] ? 0 : 1;     // own values stay
    index_set_rep_cuts(s, in, mask.data());
    return 0;
    API_CATCH_CTX(s->ctx, 2)
}

extern "C" int skder_amd_screen_rows(skder_sketches_t *s, uint32_t row_begin, uint32_t row_stride, double screen_pct,
                                     const uint32_t **ref, const uint32_t **query, uint64_t *n_pairs)
{
    if (!s) return 1;
    API_TRY
    HIPCHECK(hipSetDevice(s->ctx->device));
    if (!s->indexed && !s->index_pending) throw SkError("screen_rows: index the set first (skder_amd_sketches_index or _index_part)");
    screen_rows_impl(s, row_begin, row_stride, screen_pct, s->ctx->pairs_ref, s->ctx->pairs_query);
    if (ref) *ref = s->ctx->pairs_ref.data();
    if (query) *query = s->ctx->pairs_query.data();
    if (n_pairs) *n_pairs = s->ctx->pairs_ref.size();
    return 0;
    API_CATCH_CTX(s->ctx, 2)
}

extern "C" int skder_amd_pairs_probed(skder_sketches_t *refs, skder_sketches_t *queries, const uint32_t *ref, const uint32_t *query,
                                      uint64_t n_pairs, uint32_t *probed, uint8_t *probed_is_query)
{
    if (!refs || !queries || (n_pairs && (!ref || !query || !probed))) return 1;
    API_TRY
    settle_index(refs);
    settle_index(queries);
    pairs_probed_impl(refs, queries, ref, query, n_pairs, probed, probed_is_query);
    return 0;
    API_CATCH_CTX(refs->ctx, 2)
}

extern "C" int skder_amd_chain_pairs(skder_sketches_t *refs, skder_sketches_t *queries, const uint32_t *ref, const uint32_t *query,
                                     uint64_t n_pairs, const skder_edge_t **edges, uint64_t *n_edges)
{
    if (!refs || !queries || refs->ctx != queries->ctx || (n_pairs && (!ref || !query))) return 1;
    API_TRY
    HIPCHECK(hipSetDevice(refs->ctx->device));
    settle_index(refs);
    settle_index(queries);
    chain_pairs_impl(refs, queries, ref, query, n_pairs);
    if (edges) *edges = refs->ctx->edges.data();
    if (n_edges) *n_edges = refs->ctx->edges.size();
    return 0;
    API_CATCH_CTX(refs->ctx, 2)
}

extern "C" int skder_amd_rectangle(skder_sketches_t *refs, skder_sketches_t *queries, double screen_pct,
                                   const skder_edge_t **edges, uint64_t *n_edges)
{
    if (!refs || !queries || refs->ctx != queries->ctx) return 1;
    API_TRY
    HIPCHECK(hipSetDevice(refs->ctx->device));
    rectangle_impl(refs, queries, screen_pct);
    if (edges) *edges = refs->ctx->edges.data();
    if (n_edges) *n_edges = refs->ctx->edges.size();
    return 0;
    API_CATCH_CTX(refs->ctx, 2)
}

// ---------------------------------------------------------------------------------------------
// synthetic genomes on the device

struct SynthTile { uint64_t base_off; uint32_t genome, gpos0, npos, pad; };

__global__ __launch_bounds__(256) void synth_fill_kernel(uint8_t *__restrict__ bases, const SynthTile *__restrict__ tiles,
                                                         const uint64_t *__restrict__ lineage, const uint32_t *__restrict__ params)
{
    const SynthTile t = tiles[blockIdx.x];
    const uint64_t sp = lineage[3 * t.genome], stn = lineage[3 * t.genome + 1], iso = lineage[3 * t.genome + 2];
    const uint32_t acc = params[4 * t.genome], sppm = params[4 * t.genome + 1], ippm = params[4 * t.genome + 2];
    const uint32_t padded = (t.npos + 31u) & ~31u;
    for (uint32_t p4 = threadIdx.x * 4; p4 < padded; p4 += 256 * 4) {
        uint32_t word = 0;
        for (int k = 0; k < 4; k++) {
            uint32_t p = p4 + k;
            uint32_t ch = 'A';
            if (p < t.npos) ch = "ACGT"[synth_base(sp, stn, iso, acc, sppm, ippm, (uint64_t)t.gpos0 + p)];
            word |= ch << (8 * k);
        }
        *reinterpret_cast<uint32_t *>(bases + t.base_off + p4) = word;
    }
}

extern "C" int skder_amd_synth_fill(skder_ctx_t *ctx, uint8_t *d_bases, const skder_batch_t *b, const uint64_t *lineage,
                                    const uint32_t *params)
{
    if (!ctx || !b) return 1;
    API_TRY
    HIPCHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    std::vector<SynthTile> tiles;
    for (uint32_t g = 0; g < b->n_genomes; g++) {
        uint32_t gpos = 0;
        for (uint32_t r = b->genome_rec_begin[g]; r < b->genome_rec_begin[g + 1]; r++) {
            for (uint32_t p = 0; p < b->rec_len[r]; p += SKDER_TILE) {
                SynthTile t;
                t.base_off = b->rec_off[r] + p; t.genome = g; t.gpos0 = gpos + p;
                t.npos = b->rec_len[r] - p < SKDER_TILE ? b->rec_len[r] - p : SKDER_TILE; t.pad = 0;
                tiles.push_back(t);
            }
            gpos += b->rec_len[r];
        }
    }
    DevBuf<SynthTile> d_tiles;
    DevBuf<uint64_t> d_lin;
    DevBuf<uint32_t> d_par;
    d_tiles.resize(tiles.size(), st); d_lin.resize(3ull * b->n_genomes, st); d_par.resize(4ull * b->n_genomes, st);
    HIPCHECK(hipMemcpyAsync(d_tiles.p, tiles.data(), tiles.size() * sizeof(SynthTile), hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(d_lin.p, lineage, 3ull * b->n_genomes * 8, hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(d_par.p, params, 4ull * b->n_genomes * 4, hipMemcpyHostToDevice, st));
    if (!tiles.empty())
        hipLaunchKernelGGL(synth_fill_kernel, dim3((unsigned)tiles.size()), dim3(256), 0, st, d_bases, d_tiles.p, d_lin.p, d_par.p);
    HIPCHECK(hipStreamSynchronize(st));
    return 0;
    API_CATCH_CTX(ctx, 2)
}

// ---------------------------------------------------------------------------------------------
// drop-in entry points

// a further GPU of a multi-GPU database: the raw sketches of ALL genomes, the bucket index of the genomes it owns
struct Replica { skder_ctx *ctx = nullptr; skder_sketches *refs = nullptr; };

struct skder_db {
    skder_ctx *ctx = nullptr;
    skder_sketches *refs = nullptr;          // genomes in LISTING order, indexed (multi-GPU: GPU 0's replica)
    std::vector<Replica> more;               // GPUs 1 .. n-1 (empty: one GPU)
    GenomeNames names;
    std::vector<std::pair<std::string, uint32_t>> by_path;   // sorted (path, index)
    std::vector<skder_edge_t> rows;          // last table handed out in memory
};

extern "C" int skder_amd_parse_skani_params(const char *params, double *screen_pct, char *err, size_t errlen)
{
    std::string s = params ? params : "";
    std::vector<std::string> tok;
    size_t i = 0;
    while (i < s.size()) {
        while (i < s.size() && isspace((unsigned char)s[i])) i++;
        size_t j = i;
        while (j < s.size() && !isspace((unsigned char)s[j])) j++;
        if (j > i) tok.push_back(s.substr(i, j - i));
        i = j;
    }
    for (size_t t = 0; t < tok.size(); t++) {
        if (tok[t] == "-s") {
            if (t + 1 >= tok.size()) { set_err(err, errlen, "skani parameter -s needs a value"); return 1; }
            char *end = nullptr;
            double v = strtod(tok[t + 1].c_str(), &end);
            if (!end || *end) { set_err(err, errlen, "skani parameter -s: not a number: " + tok[t + 1]); return 1; }
            if (screen_pct) *screen_pct = v;
            t++;
        } else {
            set_err(err, errlen, "unsupported skani parameter '" + tok[t] + "': this engine implements skani's defaults and -s only");
            return 1;
        }
    }
    return 0;
}

static void db_finish(skder_db *db)
{
    index_impl(db->refs);
    db->by_path.clear();
    for (uint32_t i = 0; i < db->names.path.size(); i++) db->by_path.emplace_back(db->names.path[i], i);
    std::stable_sort(db->by_path.begin(), db->by_path.end());
}

static void db_destroy(skder_db *db)
{
    if (!db) return;
    for (Replica &r : db->more) { skder_amd_sketches_free(r.refs); skder_amd_ctx_destroy(r.ctx); }
    skder_amd_sketches_free(db->refs);
    skder_amd_ctx_destroy(db->ctx);
    delete db;
}

// ---- several GPUs in one process: one host thread per GPU for every stage
static uint32_t db_gpus(const skder_db *db) { return 1u + (uint32_t)db->more.size(); }
static skder_sketches *db_refs(skder_db *db, uint32_t d) { return d == 0 ? db->refs : db->more[d - 1].refs; }

template <typename F>
static void per_gpu(uint32_t n, F fn)
{
    std::vector<std::string> errs(n);
    std::vector<std::thread> th;
    auto run = [&](uint32_t d) {
        try { fn(d); } catch (const std::exception &e) { errs[d] = e.what()[0] ? e.what() : "error"; }
    };
    for (uint32_t d = 1; d < n; d++) th.emplace_back(run, d);
    run(0);
    for (auto &t : th) t.join();
    for (uint32_t d = 0; d < n; d++)
        if (!errs[d].empty()) throw SkError("GPU " + std::to_string(d) + ": " + errs[d]);
}

// pairs to the GPU that owns (index mod n) the genome they probe; `probe_side`: the set the routing looks at
static void route_by_probed(skder_sketches *SA, skder_sketches *SB, const std::vector<uint32_t> &ref, const std::vector<uint32_t> &query,
                            uint32_t n, bool by_ref_only, std::vector<std::vector<uint32_t>> &oref, std::vector<std::vector<uint32_t>> &oquery)
{
    oref.assign(n, {}); oquery.assign(n, {});
    std::vector<uint32_t> probed(ref.size());
    if (!ref.empty() && !by_ref_only) pairs_probed_impl(SA, SB, ref.data(), query.data(), ref.size(), probed.data(), nullptr);
    for (size_t p = 0; p < ref.size(); p++) {
        const uint32_t d = (by_ref_only ? ref[p] : probed[p]) % n;
        oref[d].push_back(ref[p]); oquery[d].push_back(query[p]);
    }
}

// all-pairs edges of a multi-GPU database: rows screened in shares, pairs chained by the owner of the probed genome
static void db_triangle_edges_multi(skder_db *db, double screen_pct, std::vector<skder_edge_t> &E)
{
    const uint32_t n = db_gpus(db);
    std::vector<std::vector<uint32_t>> sref(n), squery(n);
    per_gpu(n, [&](uint32_t d) {
        skder_sketches *s = db_refs(db, d);
        HIPCHECK(hipSetDevice(s->ctx->device));
        screen_rows_impl(s, d, n, screen_pct, sref[d], squery[d]);
    });
    std::vector<uint32_t> ref, query;
    for (uint32_t d = 0; d < n; d++) { ref.insert(ref.end(), sref[d].begin(), sref[d].end()); query.insert(query.end(), squery[d].begin(), squery[d].end()); }
    std::vector<std::vector<uint32_t>> oref, oquery;
    route_by_probed(db->refs, db->refs, ref, query, n, false, oref, oquery);
    per_gpu(n, [&](uint32_t d) {
        skder_sketches *s = db_refs(db, d);
        HIPCHECK(hipSetDevice(s->ctx->device));
        chain_pairs_impl(s, s, oref[d].data(), oquery[d].data(), oref[d].size());
    });
    E.clear();
    for (uint32_t d = 0; d < n; d++) { const auto &e = db_refs(db, d)->ctx->edges; E.insert(E.end(), e.begin(), e.end()); }
}

static double wall_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

extern "C" skder_db_t *skder_amd_sketch_n50(const char *listing, int device, const char *n50_tsv, char *err, size_t errlen)
{
    if (!listing) { set_err(err, errlen, "null argument"); return nullptr; }
    const double t0 = wall_ms();
    skder_ctx *ctx = skder_amd_ctx_create(device, err, errlen);
    if (!ctx) return nullptr;
    skder_db *db = new skder_db();
    db->ctx = ctx;
    try {
        const double t1 = wall_ms();
        std::vector<std::string> paths = read_listing(listing);
        db->refs = skder_amd_sketches_new(ctx);
        sketch_files(db->refs, paths, db->names);
        const double t2 = wall_ms();
        db_finish(db);
        if (n50_tsv) write_n50_tsv(n50_tsv, db->names);
        if (getenv("SKDER_AMD_DEBUG"))
            fprintf(stderr, "[skder_amd] database of %zu files: context %.1f ms, listing + ingest + sketch %.1f ms, index + N50 table %.1f ms\n", paths.size(),
                    t1 - t0, t2 - t1, wall_ms() - t2);
        return db;
    } catch (const std::exception &e) {
        set_err(err, errlen, e.what());
        db_destroy(db);
        return nullptr;
    }
}

extern "C" skder_db_t *skder_amd_db_from_sketches(skder_sketches_t *src, int device, uint32_t n_names, const char *const *paths,
                                                  const char *const *first_names, const uint64_t *n50, char *err, size_t errlen)
{
    if (!src || !paths) { set_err(err, errlen, "null argument"); return nullptr; }
    if (n_names != src->n_genomes) { set_err(err, errlen, "paths / first_names / n50 must hold one entry per genome of the sketch set"); return nullptr; }
    skder_ctx *ctx = skder_amd_ctx_create(device, err, errlen);
    if (!ctx) return nullptr;
    skder_db *db = new skder_db();
    db->ctx = ctx;
    try {
        HIPCHECK(hipSetDevice(src->ctx->device));
        HIPCHECK(hipStreamSynchronize(src->ctx->stream));          // the sketches are complete before another context reads them
        skder_raw_view_t v;
        if (skder_amd_sketches_view(src, &v) != 0) throw SkError("sketches_view failed");
        HIPCHECK(hipSetDevice(ctx->device));
        db->refs = skder_amd_sketches_new(ctx);
        if (v.n_genomes && skder_amd_sketches_append_raw(db->refs, &v) != 0) throw SkError(ctx->last_error);
        for (uint32_t g = 0; g < v.n_genomes; g++) {
            if (!paths[g]) throw SkError("null path");
            db->names.path.emplace_back(paths[g]);
            db->names.first_name.emplace_back(first_names && first_names[g] ? first_names[g] : "");
            db->names.n50.push_back(n50 ? n50[g] : 0);
        }
        db_finish(db);
        return db;
    } catch (const std::exception &e) {
        set_err(err, errlen, e.what());
        db_destroy(db);
        return nullptr;
    }
}

// Peer access from `dev` to every other device of the list (once per process and pair): with it a copy between two GPUs goes
// over their xGMI link directly; without it the runtime stages it through host memory (still correct, an order of magnitude
// slower) -- said on stderr under SKDER_AMD_DEBUG, never silently.
static std::atomic<uint32_t> g_peer_fallbacks{0};      // device pairs (ordered) whose copies go through host memory
extern "C" uint32_t skder_amd_peer_fallbacks() { return g_peer_fallbacks.load(); }

extern std::atomic<int> g_ani_output_raw;      // chain.hip
extern "C" int skder_amd_set_ani_output(int raw)
{
    if (raw != 0 && raw != 1) return -1;
    return g_ani_output_raw.exchange(raw);
}
static void enable_peer_access(int dev, const std::vector<int> &devices)
{
    HIPCHECK(hipSetDevice(dev));
    for (int other : devices) {
        if (other == dev) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, dev, other) != hipSuccess) can = 0;
        if (can) {
            const hipError_t e = hipDeviceEnablePeerAccess(other, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) can = 0;
            (void)hipGetLastError();
        }
        if (!can) g_peer_fallbacks++;
        if (!can && getenv("SKDER_AMD_DEBUG"))
            fprintf(stderr, "[skder_amd] GPU %d has no peer access to GPU %d: sketches from there are copied through host memory\n", dev, other);
    }
}

// The "all-gather" of one GPU: the raw sketches of every share, in listing order, into `dst` (not indexed yet).  xGMI is
// point to point -- seven links per GPU, one per peer -- so the pulls from the n - 1 peers are issued on n - 1 STREAMS and run
// side by side, each over its own link; issued one after the other on one stream they would take (n - 1) x one link's time.
static void gather_raw_views(skder_sketches *dst, const std::vector<skder_raw_view_t> &views)
{
    if (dst->indexed || dst->index_pending || dst->n_genomes) throw SkError("gather_raw_views: the set must be empty");
    skder_ctx *ctx = dst->ctx;
    HIPCHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    uint64_t ns = 0, nm = 0;
    for (const auto &v : views) { ns += v.n_seeds; nm += v.n_markers; }
    dst->seed_kmer.resize(ns, st);
    dst->seed_gpos.reserve(ns + 32, 0, st);
    dst->seed_gpos.resize(ns, st);
    dst->seed_ctg.resize(ns, st);
    dst->markers.resize(nm, st);
    HIPCHECK(hipStreamSynchronize(st));
    std::vector<hipStream_t> streams(views.size(), nullptr);
    struct Cleanup { std::vector<hipStream_t> &s; ~Cleanup() { for (auto x : s) if (x) (void)hipStreamDestroy(x); } } cleanup{streams};
    uint64_t so = 0, mo = 0;
    for (size_t e = 0; e < views.size(); e++) {
        const skder_raw_view_t &v = views[e];
        HIPCHECK(hipStreamCreateWithFlags(&streams[e], hipStreamNonBlocking));
        if (v.n_seeds) {
            HIPCHECK(hipMemcpyAsync(dst->seed_kmer.p + so, v.d_seed_kmer, v.n_seeds * 4, hipMemcpyDefault, streams[e]));
            HIPCHECK(hipMemcpyAsync(dst->seed_gpos.p + so, v.d_seed_gpos, v.n_seeds * 4, hipMemcpyDefault, streams[e]));
            HIPCHECK(hipMemcpyAsync(dst->seed_ctg.p + so, v.d_seed_ctg, v.n_seeds * 4, hipMemcpyDefault, streams[e]));
        }
        if (v.n_markers) HIPCHECK(hipMemcpyAsync(dst->markers.p + mo, v.d_markers, v.n_markers * 8, hipMemcpyDefault, streams[e]));
        size_t rg = 0;
        for (uint32_t g = 0; g < v.n_genomes; g++) {
            dst->h_seed_off.push_back(so + v.h_seed_off[g + 1] - v.h_seed_off[0]);
            dst->h_marker_off.push_back(mo + v.h_marker_off[g + 1] - v.h_marker_off[0]);
            dst->h_genome_len.push_back(v.h_genome_len[g]);
            dst->h_genome_nrec.push_back(v.h_genome_nrec[g]);
            for (uint32_t r = 0; r <= v.h_genome_nrec[g]; r++) dst->h_rec_goff.push_back(v.h_rec_goff[rg++]);
        }
        dst->n_genomes += v.n_genomes;
        so += v.n_seeds; mo += v.n_markers;
    }
    for (auto x : streams) HIPCHECK(hipStreamSynchronize(x));
}

extern "C" skder_db_t *skder_amd_sketch_multi(const char *listing, const int *devices, int n_devices, const char *n50_tsv, char *err,
                                              size_t errlen)
{
    if (!listing || !devices || n_devices < 1) { set_err(err, errlen, "null argument"); return nullptr; }
    if (n_devices == 1) return skder_amd_sketch_n50(listing, devices[0], n50_tsv, err, errlen);
    const uint32_t n = (uint32_t)n_devices;
    skder_db *db = new skder_db();
    std::vector<skder_sketches *> part(n, nullptr);
    try {
        std::vector<std::string> paths = read_listing(listing);
        const size_t G = paths.size();
        // one context per GPU; GPU d reads and sketches the d-th contiguous share of the listing
        db->more.resize(n - 1);
        for (uint32_t d = 0; d < n; d++) {
            skder_ctx *c = skder_amd_ctx_create(devices[d], err, errlen);
            if (!c) throw SkError(std::string("cannot open GPU ") + std::to_string(devices[d]) + ": " + (err ? err : ""));
            if (d == 0) db->ctx = c; else db->more[d - 1].ctx = c;
        }
        auto ctx_of = [&](uint32_t d) { return d == 0 ? db->ctx : db->more[d - 1].ctx; };
        {
            std::vector<int> devs(devices, devices + n);
            for (uint32_t d = 0; d < n; d++) enable_peer_access(devices[d], devs);
        }
        std::vector<GenomeNames> names(n);
        per_gpu(n, [&](uint32_t d) {
            HIPCHECK(hipSetDevice(ctx_of(d)->device));
            part[d] = skder_amd_sketches_new(ctx_of(d));
            const size_t lo = G * d / n, hi = G * (d + 1) / n;
            sketch_files(part[d], std::vector<std::string>(paths.begin() + lo, paths.begin() + hi), names[d], std::max(4u, ingest_threads() / n));
        });
        for (uint32_t d = 0; d < n; d++) {
            db->names.path.insert(db->names.path.end(), names[d].path.begin(), names[d].path.end());
            db->names.first_name.insert(db->names.first_name.end(), names[d].first_name.begin(), names[d].first_name.end());
            db->names.n50.insert(db->names.n50.end(), names[d].n50.begin(), names[d].n50.end());
        }
        // "all-gather": every GPU pulls the raw sketches of every share (peer copies over xGMI, one stream per source so that all
        // of a GPU's links carry traffic at once), in listing order; then the bucket index of the genomes it owns (index mod n)
        // and everybody's chunk tables
        std::vector<skder_raw_view_t> views(n);
        for (uint32_t d = 0; d < n; d++) if (skder_amd_sketches_view(part[d], &views[d]) != 0) throw SkError("sketches_view failed");
        per_gpu(n, [&](uint32_t d) {
            HIPCHECK(hipSetDevice(ctx_of(d)->device));
            skder_sketches *all = skder_amd_sketches_new(ctx_of(d));
            if (d == 0) db->refs = all; else db->more[d - 1].refs = all;
            gather_raw_views(all, views);
            std::vector<uint8_t> own(G);
            for (size_t g = 0; g < G; g++) own[g] = (g % n == d) ? 1 : 0;
            index_begin(all, ctx_of(d)->stream, own.data());
            index_finish(all);
        });
        for (uint32_t d = 0; d < n; d++) { skder_amd_sketches_free(part[d]); part[d] = nullptr; }
        // repetitive-k-mer cut-offs: each genome's from its owner
        std::vector<uint32_t> rep(G, 0xFFFFFFFFu);
        for (size_t g = 0; g < G; g++) rep[g] = db_refs(db, (uint32_t)(g % n))->h_meta[g].rep_cut;
        per_gpu(n, [&](uint32_t d) {
            HIPCHECK(hipSetDevice(ctx_of(d)->device));
            std::vector<uint8_t> mask(G);
            for (size_t g = 0; g < G; g++) mask[g] = (g % n == d) ? 0 : 1;
            index_set_rep_cuts(db_refs(db, d), rep.data(), mask.data());
        });
        db->by_path.clear();
        for (uint32_t i = 0; i < db->names.path.size(); i++) db->by_path.emplace_back(db->names.path[i], i);
        std::stable_sort(db->by_path.begin(), db->by_path.end());
        if (n50_tsv) write_n50_tsv(n50_tsv, db->names);
        return db;
    } catch (const std::exception &e) {
        set_err(err, errlen, e.what());
        for (auto *p : part) skder_amd_sketches_free(p);
        db_destroy(db);
        return nullptr;
    }
}

extern "C" int skder_amd_triangle_multi(const char *listing, double min_af_pct, double screen_pct, const int *devices, int n_devices,
                                        const char *out_tsv, const char *n50_tsv, char *err, size_t errlen)
{
    if (!listing || !out_tsv) { set_err(err, errlen, "null argument"); return 1; }
    skder_db *db = skder_amd_sketch_multi(listing, devices, n_devices, n50_tsv, err, errlen);
    if (!db) return 1;
    int rc = skder_amd_db_triangle(db, min_af_pct, screen_pct, out_tsv, nullptr, nullptr, err, errlen);
    db_destroy(db);
    return rc;
}

extern "C" skder_db_t *skder_amd_sketch(const char *listing, int device, char *err, size_t errlen)
{
    return skder_amd_sketch_n50(listing, device, nullptr, err, errlen);
}

extern "C" void skder_amd_db_free(skder_db_t *db) { db_destroy(db); }
extern "C" uint32_t skder_amd_db_size(skder_db_t *db) { return db ? (uint32_t)db->names.path.size() : 0; }
extern "C" const char *skder_amd_db_path(skder_db_t *db, uint32_t i)
{
    return db && i < db->names.path.size() ? db->names.path[i].c_str() : nullptr;
}
extern "C" uint64_t skder_amd_db_n50(skder_db_t *db, uint32_t i) { return db && i < db->names.n50.size() ? db->names.n50[i] : 0; }

// All-pairs table of the database in skani's triangle conventions.  skani numbers genomes by
// ascending path whereas the database keeps listing order: rows are formed in rank space
// (Ref = the path that sorts first, AFs swapped along) and mapped back to listing indices.
static void db_triangle_rows(skder_db *db, double min_af_pct, double screen_pct)
{
    const uint32_t n = (uint32_t)db->names.path.size();
    std::vector<uint32_t> rank(n), perm(n);
    for (uint32_t r = 0; r < n; r++) { perm[r] = db->by_path[r].second; rank[perm[r]] = r; }
    // ONE copy of the edge list from here on: the engine's buffer is taken over (swap), ordered and filtered in place, and
    // becomes db->rows
    std::vector<skder_edge_t> &E = db->rows;
    E.clear();
    if (db->more.empty()) { triangle_rows_impl(db->refs, 0, 1, screen_pct); E.swap(db->ctx->edges); }
    else db_triangle_edges_multi(db, screen_pct, E);
    // (both passes over the list on the host threads: at 10^8 records a plain loop is a second per pass)
    host_parallel_chunks(E.size(), [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; i++) {
            skder_edge_t &e = E[i];
            uint32_t a = rank[e.ref], b = rank[e.query];
            if (a > b) { std::swap(a, b); std::swap(e.af_ref, e.af_query); }
            e.ref = a; e.query = b;
        }
    });
    triangle_rows_order_inplace(E, min_af_pct);
    host_parallel_chunks(E.size(), [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; i++) { E[i].ref = perm[E[i].ref]; E[i].query = perm[E[i].query]; }
    });
}

extern "C" int skder_amd_db_triangle(skder_db_t *db, double min_af_pct, double screen_pct, const char *out_tsv,
                                     const skder_edge_t **edges, uint64_t *n_edges, char *err, size_t errlen)
{
    if (!db) { set_err(err, errlen, "null argument"); return 1; }
    try {
        HIPCHECK(hipSetDevice(db->ctx->device));
        const double t0 = wall_ms();
        db_triangle_rows(db, min_af_pct, screen_pct);
        const double t1 = wall_ms();
        if (out_tsv) write_rows_tsv(out_tsv, db->rows.data(), db->rows.size(), db->names, db->names);
        if (getenv("SKDER_AMD_DEBUG"))
            fprintf(stderr, "[skder_amd] table of %zu rows: screen + chain + row order %.1f ms, text %.1f ms\n", db->rows.size(), t1 - t0, wall_ms() - t1);
        if (edges) *edges = db->rows.data();
        if (n_edges) *n_edges = db->rows.size();
        return 0;
    } catch (const std::exception &e) {
        set_err(err, errlen, e.what());
        return 2;
    }
}

extern "C" int skder_amd_triangle_n50(const char *listing, double min_af_pct, double screen_pct, int device, const char *out_tsv,
                                      const char *n50_tsv, char *err, size_t errlen)
{
    if (!listing || !out_tsv) { set_err(err, errlen, "null argument"); return 1; }
    skder_db *db = skder_amd_sketch_n50(listing, device, n50_tsv, err, errlen);
    if (!db) return 1;
    int rc = skder_amd_db_triangle(db, min_af_pct, screen_pct, out_tsv, nullptr, nullptr, err, errlen);
    db_destroy(db);
    return rc;
}

extern "C" int skder_amd_triangle(const char *listing, double min_af_pct, double screen_pct, int device, const char *out_tsv,
                                  char *err, size_t errlen)
{
    return skder_amd_triangle_n50(listing, min_af_pct, screen_pct, device, out_tsv, nullptr, err, errlen);
}

extern "C" int skder_amd_dist(const char *ref_listing, const char *query_listing, double min_af_pct, double screen_pct, int device,
                              const char *out_tsv, char *err, size_t errlen)
{
    skder_ctx *ctx = skder_amd_ctx_create(device, err, errlen);
    if (!ctx) return 1;
    skder_sketches *r = nullptr, *q = nullptr;
    int rc = 0;
    try {
        std::vector<std::string> rp = read_listing(ref_listing), qp = read_listing(query_listing);
        r = skder_amd_sketches_new(ctx);
        q = skder_amd_sketches_new(ctx);
        GenomeNames rn, qn;
        sketch_files(r, rp, rn);
        sketch_files(q, qp, qn);
        rectangle_impl(r, q, screen_pct);
        write_rect_tsv(out_tsv, ctx->edges, rn, qn, min_af_pct);
    } catch (const std::exception &e) {
        set_err(err, errlen, e.what());
        rc = 2;
    }
    skder_amd_sketches_free(r);
    skder_amd_sketches_free(q);
    skder_amd_ctx_destroy(ctx);
    return rc;
}

// append genome g of `src` (raw sketch, D2D copy) to the un-indexed set `dst`
static void append_genome(skder_sketches *dst, skder_sketches *src, uint32_t g)
{
    skder_raw_view_t v;
    skder_amd_sketches_view(src, &v);
    skder_raw_view_t one = v;
    uint64_t so[2] = {v.h_seed_off[g], v.h_seed_off[g + 1]}, mo[2] = {v.h_marker_off[g], v.h_marker_off[g + 1]};
    size_t rg = 0;
    for (uint32_t i = 0; i < g; i++) rg += v.h_genome_nrec[i] + 1;
    one.n_genomes = 1;
    one.n_seeds = so[1] - so[0]; one.n_markers = mo[1] - mo[0];
    one.d_seed_kmer = v.d_seed_kmer + so[0]; one.d_seed_gpos = v.d_seed_gpos + so[0]; one.d_seed_ctg = v.d_seed_ctg + so[0];
    one.d_markers = v.d_markers + mo[0];
    one.h_seed_off = so; one.h_marker_off = mo;
    one.h_genome_len = v.h_genome_len + g; one.h_genome_nrec = v.h_genome_nrec + g;
    one.h_rec_goff = v.h_rec_goff + rg;
    if (skder_amd_sketches_append_raw(dst, &one) != 0) throw SkError(src->ctx->last_error);
}

extern "C" int skder_amd_search_batch(skder_db_t *db, const char *const *query_paths, uint32_t n_queries, double min_af_pct,
                                      double screen_pct, const char *const *out_tsvs, const skder_edge_t **edges,
                                      uint64_t *n_edges, char *err, size_t errlen)
{
    return skder_amd_search_batch_live(db, query_paths, n_queries, min_af_pct, screen_pct, out_tsvs, nullptr, edges, n_edges, err, errlen);
}

extern "C" int skder_amd_search_batch_live(skder_db_t *db, const char *const *query_paths, uint32_t n_queries, double min_af_pct,
                                           double screen_pct, const char *const *out_tsvs, const uint8_t *live, const skder_edge_t **edges,
                                           uint64_t *n_edges, char *err, size_t errlen)
{
    if (!db || !query_paths) { set_err(err, errlen, "null argument"); return 1; }
    skder_sketches *q = nullptr;
    int rc = 0;
    try {
        HIPCHECK(hipSetDevice(db->ctx->device));
        GenomeNames qn;
        q = skder_amd_sketches_new(db->ctx);
        for (uint32_t k = 0; k < n_queries; k++) {
            if (!query_paths[k]) throw SkError("null query path");
            const std::string qp(query_paths[k]);
            auto it = std::lower_bound(db->by_path.begin(), db->by_path.end(), std::make_pair(qp, 0u));
            if (it != db->by_path.end() && it->first == qp) {     // resident: reuse its sketch
                append_genome(q, db->refs, it->second);
                qn.path.push_back(db->names.path[it->second]);
                qn.first_name.push_back(db->names.first_name[it->second]);
                qn.n50.push_back(db->names.n50[it->second]);
            } else {
                sketch_files(q, {qp}, qn);
            }
        }
        db->rows.clear();
        if (n_queries && db->more.empty()) {
            rectangle_impl(db->refs, q, screen_pct, live);
            db->rows.swap(db->ctx->edges);
            rect_rows_order_inplace(db->rows, min_af_pct);
        } else if (n_queries) {
            // several GPUs: every GPU gets the queries' sketches (a peer copy of GPU 0's), screens a share of the queries
            // against all markers and chains the pairs whose DATABASE genome it owns -- if the pair probes that genome its
            // index is there, if it probes the query every GPU has indexed the queries
            const uint32_t n = db_gpus(db);
            skder_raw_view_t qv;
            if (skder_amd_sketches_view(q, &qv) != 0) throw SkError("sketches_view failed");
            std::vector<skder_sketches *> qs(n, nullptr);
            qs[0] = q;
            std::vector<std::vector<uint32_t>> sref(n), squery(n), oref, oquery;
            try {
                per_gpu(n, [&](uint32_t d) {
                    skder_sketches *refs = db_refs(db, d);
                    HIPCHECK(hipSetDevice(refs->ctx->device));
                    if (d) {
                        qs[d] = skder_amd_sketches_new(refs->ctx);
                        if (skder_amd_sketches_append_raw(qs[d], &qv) != 0) throw SkError(refs->ctx->last_error);
                    }
                    index_impl(qs[d]);
                    std::vector<uint32_t> rows;
                    for (uint32_t k = d; k < n_queries; k += n) rows.push_back(k);
                    if (!rows.empty()) screen_pairs(refs, qs[d], rows, false, screen_pct, squery[d], sref[d]);
                });
                std::vector<uint32_t> ref, query;
                for (uint32_t d = 0; d < n; d++)
                    for (size_t k = 0; k < sref[d].size(); k++)
                        if (!live || live[sref[d][k]]) { ref.push_back(sref[d][k]); query.push_back(squery[d][k]); }
                route_by_probed(db->refs, q, ref, query, n, true, oref, oquery);
                per_gpu(n, [&](uint32_t d) {
                    skder_sketches *refs = db_refs(db, d);
                    HIPCHECK(hipSetDevice(refs->ctx->device));
                    chain_pairs_impl(refs, qs[d], oref[d].data(), oquery[d].data(), oref[d].size());
                });
            } catch (...) {
                for (uint32_t d = 1; d < n; d++) skder_amd_sketches_free(qs[d]);
                throw;
            }
            std::vector<skder_edge_t> E;
            for (uint32_t d = 0; d < n; d++) { const auto &e = db_refs(db, d)->ctx->edges; E.insert(E.end(), e.begin(), e.end()); }
            for (uint32_t d = 1; d < n; d++) skder_amd_sketches_free(qs[d]);
            db->rows.swap(E);
            rect_rows_order_inplace(db->rows, min_af_pct);
        }
        if (out_tsvs) {
            size_t lo = 0;
            for (uint32_t k = 0; k < n_queries; k++) {
                size_t hi = lo;
                while (hi < db->rows.size() && db->rows[hi].query == k) hi++;
                if (out_tsvs[k]) write_rows_tsv(out_tsvs[k], db->rows.data() + lo, hi - lo, db->names, qn);
                lo = hi;
            }
        }
        if (edges) *edges = db->rows.data();
        if (n_edges) *n_edges = db->rows.size();
    } catch (const std::exception &e) {
        set_err(err, errlen, e.what());
        rc = 2;
    }
    skder_amd_sketches_free(q);
    return rc;
}

extern "C" int skder_amd_search(skder_db_t *db, const char *query_path, double min_af_pct, double screen_pct, const char *out_tsv,
                                char *err, size_t errlen)
{
    if (!db || !query_path || !out_tsv) { set_err(err, errlen, "null argument"); return 1; }
    return skder_amd_search_batch(db, &query_path, 1, min_af_pct, screen_pct, &out_tsv, nullptr, nullptr, err, errlen);
}

extern "C" int skder_amd_db_save(skder_db_t *db, const char *store_path, char *err, size_t errlen)
{
    if (!db || !store_path) { set_err(err, errlen, "null argument"); return 1; }
    try {
        HIPCHECK(hipSetDevice(db->ctx->device));
        store_save(store_path, db->refs, db->names);
        return 0;
    } catch (const std::exception &e) {
        set_err(err, errlen, e.what());
        return 2;
    }
}

extern "C" skder_db_t *skder_amd_db_load(const char *store_path, int device, char *err, size_t errlen)
{
    if (!store_path) { set_err(err, errlen, "null argument"); return nullptr; }
    skder_ctx *ctx = skder_amd_ctx_create(device, err, errlen);
    if (!ctx) return nullptr;
    skder_db *db = new skder_db();
    db->ctx = ctx;
    try {
        db->refs = skder_amd_sketches_new(ctx);
        store_load(store_path, db->refs, db->names);
        db_finish(db);
        return db;
    } catch (const std::exception &e) {
        set_err(err, errlen, e.what());
        db_destroy(db);
        return nullptr;
    }
}
