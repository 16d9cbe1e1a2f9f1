// host_io.hip -- host side of the drop-in entry points: FASTA/gzip ingest, listing files and the
// skani-format edge table (columns, %.2f, names, filter and row order of SURVEY.md 8c V1-V6).
// No compute happens here: bases go to HBM unmodified (1 byte per base) and the kernels do the rest.
#include "host_io.h"

#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <thread>

void read_fasta(const std::string &path, HostGenome &g)
{
    g.path = path;
    g.first_name.clear(); g.rec_len.clear(); g.bases.clear();
    gzFile f = gzopen(path.c_str(), "rb");
    if (!f) throw SkError("cannot open " + path);
    gzbuffer(f, 1 << 20);
    std::vector<uint64_t> all_len;
    std::vector<char> buf(1 << 20);
    std::string cur_name;
    bool in_header = false, have_rec = false, have_first = false;
    size_t rec_start = 0;   // offset in g.bases where the current record starts
    auto close_rec = [&]() {
        if (!have_rec) return;
        size_t len = g.bases.size() - rec_start;
        all_len.push_back(len);
        if (len >= ANI_MIN_CONTIG) {
            if (len > 0x7FFFFFFFull) throw SkError("record longer than 2^31 in " + path);
            g.rec_len.push_back((uint32_t)len);
            if (!have_first) { g.first_name = cur_name; have_first = true; }
        } else {
            g.bases.resize(rec_start);   // records below 500 bp are ignored entirely
        }
    };
    bool at_line_start = true;
    for (;;) {
        int n = gzread(f, buf.data(), (unsigned)buf.size());
        if (n < 0) { gzclose(f); throw SkError("read error in " + path); }
        if (n == 0) break;
        for (int i = 0; i < n; i++) {
            char c = buf[i];
            if (in_header) {
                if (c == '\n') { in_header = false; at_line_start = true; }
                else if (c != '\r') cur_name.push_back(c);
                continue;
            }
            if (c == '\n') { at_line_start = true; continue; }
            if (at_line_start && c == '>') {
                close_rec();
                have_rec = true;
                rec_start = g.bases.size();
                cur_name.clear();
                in_header = true;
                continue;
            }
            at_line_start = false;
            if (c == '\r' || c == ' ' || c == '\t') continue;
            if (have_rec) g.bases.push_back((uint8_t)c);
        }
    }
    close_rec();
    gzclose(f);
    if (all_len.empty()) throw SkError("no FASTA records in " + path);
    // N50 as util.py:686-724
    std::sort(all_len.begin(), all_len.end());
    uint64_t tot = 0;
    for (uint64_t l : all_len) tot += l;
    uint64_t half = tot / 2, cum = 0;
    g.n50 = all_len[0];
    for (size_t i = all_len.size(); i-- > 0;) {
        cum += all_len[i];
        if (cum >= half) { g.n50 = all_len[i]; break; }
    }
}

std::vector<std::string> read_listing(const std::string &path)
{
    FILE *f = fopen(path.c_str(), "r");
    if (!f) throw SkError("cannot open listing " + path);
    std::vector<std::string> v;
    char *line = nullptr;
    size_t cap = 0;
    ssize_t len;
    while ((len = getline(&line, &cap, f)) >= 0) {
        while (len && (line[len - 1] == '\n' || line[len - 1] == '\r' || line[len - 1] == ' ')) line[--len] = 0;
        if (len) v.emplace_back(line);
    }
    free(line);
    fclose(f);
    return v;
}

void sketch_files(skder_sketches *s, const std::vector<std::string> &paths, GenomeNames &names)
{
    skder_ctx *ctx = s->ctx;
    hipStream_t st = ctx->stream;
    const size_t batch_bytes = 1ull << 30;
    size_t i0 = 0;
    unsigned nthreads = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    while (i0 < paths.size()) {
        // read files in parallel until the batch holds ~1 GB of bases
        std::vector<HostGenome> gs;
        size_t i1 = i0, bytes = 0;
        while (i1 < paths.size() && bytes < batch_bytes) {
            size_t chunk = std::min<size_t>(paths.size() - i1, 64);
            size_t base = gs.size();
            gs.resize(base + chunk);
            std::atomic<size_t> next(0);
            std::string first_err;
            std::atomic<bool> failed(false);
            std::vector<std::thread> th;
            for (unsigned t = 0; t < nthreads; t++)
                th.emplace_back([&]() {
                    for (;;) {
                        size_t k = next.fetch_add(1);
                        if (k >= chunk) break;
                        try { read_fasta(paths[i1 + k], gs[base + k]); }
                        catch (const std::exception &e) {
                            if (!failed.exchange(true)) first_err = e.what();
                        }
                    }
                });
            for (auto &t : th) t.join();
            if (failed) throw SkError(first_err);
            for (size_t k = 0; k < chunk; k++) bytes += gs[base + k].bases.size();
            i1 += chunk;
        }
        // device layout: records 32-B aligned, 32 B in front, SKDER_TILE + 32 behind
        std::vector<uint64_t> rec_off;
        std::vector<uint32_t> rec_len, gbegin;
        uint64_t off = 32;
        for (auto &g : gs) {
            gbegin.push_back((uint32_t)rec_len.size());
            for (uint32_t l : g.rec_len) {
                rec_off.push_back(off);
                rec_len.push_back(l);
                off += (l + 31ull) & ~31ull;
            }
        }
        gbegin.push_back((uint32_t)rec_len.size());
        const uint64_t total = off + SKDER_TILE + 64;
        uint8_t *h = nullptr, *d = nullptr;
        HIPCHECK(hipHostMalloc(&h, total));
        memset(h, 'A', 32);
        {
            size_t r = 0;
            for (auto &g : gs) {
                size_t src = 0;
                for (uint32_t l : g.rec_len) {
                    memcpy(h + rec_off[r], g.bases.data() + src, l);
                    uint64_t padded = (l + 31ull) & ~31ull;
                    memset(h + rec_off[r] + l, 'A', padded - l);
                    src += l; r++;
                }
            }
            memset(h + off, 'A', SKDER_TILE + 64);
        }
        HIPCHECK(hipMalloc(&d, total));
        HIPCHECK(hipMemcpyAsync(d, h, total, hipMemcpyHostToDevice, st));
        skder_batch_t b;
        b.n_genomes = (uint32_t)gs.size();
        b.n_records = (uint32_t)rec_len.size();
        b.rec_off = rec_off.data(); b.rec_len = rec_len.data(); b.genome_rec_begin = gbegin.data();
        try {
            sketch_batch_impl(s, d, &b);
        } catch (...) {
            (void)hipFree(d); (void)hipHostFree(h);
            throw;
        }
        HIPCHECK(hipStreamSynchronize(st));
        (void)hipFree(d); (void)hipHostFree(h);
        for (auto &g : gs) { names.path.push_back(g.path); names.first_name.push_back(g.first_name); }
        i0 = i1;
    }
}

// ---------------------------------------------------------------------------------------------
// TSV output

static const char *TSV_HEADER = "Ref_file\tQuery_file\tANI\tAlign_fraction_ref\tAlign_fraction_query\tRef_name\tQuery_name\n";

// hashbrown (SwissTable) + FxHash iteration-order model: bucket = (key*K) & mask, tables
// grow 4 -> 8 -> 16 ... when items exceed 3, 7, 14, 28, ...; iteration ascends over buckets.
struct FxMap {
    std::vector<uint64_t> key;
    std::vector<uint8_t> full;
    uint32_t buckets = 0, items = 0;
    static uint32_t capacity(uint32_t b) { return b < 8 ? b - 1 : b / 8 * 7; }
    void place(uint64_t k) {
        uint64_t h = k * 0x517cc1b727220a95ULL;
        uint32_t mask = buckets - 1, pos = (uint32_t)(h & mask);
        while (full[pos]) pos = (pos + 1) & mask;
        full[pos] = 1; key[pos] = k;
    }
    void insert(uint64_t k) {
        if (buckets == 0 || items + 1 > capacity(buckets)) {
            FxMap n;
            n.buckets = buckets ? buckets * 2 : 4;
            n.key.assign(n.buckets, 0); n.full.assign(n.buckets, 0);
            for (uint32_t b = 0; b < buckets; b++) if (full[b]) n.place(key[b]);
            n.items = items;
            *this = n;
        }
        place(k);
        items++;
    }
};

static void print_row(FILE *o, const std::string &rf, const std::string &qf, const skder_edge_t &e, const std::string &rn,
                      const std::string &qn)
{
    // skani keeps its results in single precision and prints percentages with two decimals
    float ani = (float)e.ani, afr = (float)e.af_ref, afq = (float)e.af_query;
    fprintf(o, "%s\t%s\t%.2f\t%.2f\t%.2f\t%s\t%s\n", rf.c_str(), qf.c_str(), (double)(ani * 100.0f), (double)(afr * 100.0f),
            (double)(afq * 100.0f), rn.c_str(), qn.c_str());
}

static bool passes_min_af(const skder_edge_t &e, double min_af_pct)
{
    float afr = (float)e.af_ref, afq = (float)e.af_query;
    double mx = afr > afq ? afr : afq;
    return mx * 100.0 >= min_af_pct;   // V4: max(AF) on unrounded values
}

struct TmpFile {
    std::string tmp, final_name;
    FILE *f = nullptr;
    explicit TmpFile(const std::string &out) : final_name(out)
    {
        tmp = out + ".tmp." + std::to_string((long)getpid());
        f = fopen(tmp.c_str(), "w");
        if (!f) throw SkError("cannot write " + out);
    }
    void commit()
    {
        if (fclose(f) != 0) { f = nullptr; remove(tmp.c_str()); throw SkError("write error on " + final_name); }
        f = nullptr;
        if (rename(tmp.c_str(), final_name.c_str()) != 0) { remove(tmp.c_str()); throw SkError("cannot rename to " + final_name); }
    }
    ~TmpFile() { if (f) { fclose(f); remove(tmp.c_str()); } }
};

void write_triangle_tsv(const std::string &out, const std::vector<skder_edge_t> &edges, const GenomeNames &names, double min_af_pct)
{
    // rows by Ref index; inside a row by Query index ascending (the insertion order of skani's inner map)
    std::vector<skder_edge_t> E(edges);
    std::sort(E.begin(), E.end(), [](const skder_edge_t &a, const skder_edge_t &b) {
        return a.ref != b.ref ? a.ref < b.ref : a.query < b.query;
    });
    TmpFile tf(out);
    fputs(TSV_HEADER, tf.f);
    FxMap outer;
    for (size_t i = 0; i < E.size(); i++)
        if (i == 0 || E[i].ref != E[i - 1].ref) outer.insert(E[i].ref);
    for (uint32_t b = 0; b < outer.buckets; b++) {
        if (!outer.full[b]) continue;
        const uint32_t ref = (uint32_t)outer.key[b];
        auto lo = std::lower_bound(E.begin(), E.end(), ref, [](const skder_edge_t &e, uint32_t r) { return e.ref < r; });
        auto hi = lo;
        FxMap inner;
        while (hi != E.end() && hi->ref == ref) { inner.insert(hi->query); ++hi; }
        for (uint32_t bb = 0; bb < inner.buckets; bb++) {
            if (!inner.full[bb]) continue;
            const uint32_t q = (uint32_t)inner.key[bb];
            auto it = std::lower_bound(lo, hi, q, [](const skder_edge_t &e, uint32_t qq) { return e.query < qq; });
            if (!passes_min_af(*it, min_af_pct)) continue;
            print_row(tf.f, names.path[ref], names.path[q], *it, names.first_name[ref], names.first_name[q]);
        }
    }
    tf.commit();
}

void write_rect_tsv(const std::string &out, const std::vector<skder_edge_t> &edges, const GenomeNames &ref_names,
                    const GenomeNames &query_names, double min_af_pct)
{
    // grouped by query in listing order; references by ANI descending
    std::vector<skder_edge_t> E;
    for (const auto &e : edges) if (passes_min_af(e, min_af_pct)) E.push_back(e);
    std::sort(E.begin(), E.end(), [](const skder_edge_t &a, const skder_edge_t &b) {
        if (a.query != b.query) return a.query < b.query;
        float x = (float)a.ani, y = (float)b.ani;
        if (x != y) return x > y;
        return a.ref < b.ref;
    });
    TmpFile tf(out);
    fputs(TSV_HEADER, tf.f);
    for (const auto &e : E)
        print_row(tf.f, ref_names.path[e.ref], query_names.path[e.query], e, ref_names.first_name[e.ref],
                  query_names.first_name[e.query]);
    tf.commit();
}
