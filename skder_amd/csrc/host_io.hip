// host_io.hip -- host side of the drop-in entry points: FASTA/gzip ingest, listing files and the
// skani-format edge table (columns, %.2f, names, filter and row order of SURVEY.md 8c V1-V6).
// No compute happens here: bases go to HBM unmodified (1 byte per base) and the kernels do the rest.
#include "host_io.h"
#include "fasta.h"
#include "gunzip.h"

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <future>
#include <mutex>
#include <thread>

// growing or fixed output of the parser
namespace {
struct OutBuf {
    uint8_t *p;
    size_t n, cap;
    bool fixed;
    void need(size_t extra)
    {
        if (n + extra <= cap) return;
        if (fixed) throw SkError("region");
        size_t nc = cap * 2 > n + extra ? cap * 2 : n + extra + (1u << 20);
        uint8_t *q = static_cast<uint8_t *>(realloc(p, nc));
        if (!q) throw SkError("out of host memory");
        p = q; cap = nc;
    }
};
struct FileCloser {
    int fd = -1;
    ~FileCloser() { if (fd >= 0) close(fd); }
};
}   // namespace

static bool looks_gzip(const std::string &path)
{
    unsigned char m[2] = {0, 0};
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) throw SkError("cannot open " + path);
    const size_t k = fread(m, 1, 2, f);
    fclose(f);
    return k == 2 && m[0] == 0x1f && m[1] == 0x8b;
}

// Per-thread working memory of the reader, reused from file to file: the text buffer the FASTA scan runs over, the
// compressed-input buffer and the inflate state.  (gzopen / gzread allocate and release two 1 MB buffers per file and a
// thread_local buffer dies with its thread: with a hundred reader threads those mmap / munmap calls and the page faults
// behind them serialise on the process's address-space lock and the ingest stops scaling at 16 threads -- measured.)
struct IoScratch {
    std::vector<char> text;
    std::vector<unsigned char> zin;
    std::vector<uint8_t> zfile;        // a whole compressed file (gunzip.cpp decodes from memory)
    z_stream strm;
    bool z_ready = false;
    IoScratch() : text(1u << 20), zin(256u << 10) { memset(&strm, 0, sizeof strm); }
    ~IoScratch() { if (z_ready) inflateEnd(&strm); }
    IoScratch(const IoScratch &) = delete;
    IoScratch &operator=(const IoScratch &) = delete;
};

namespace {
// decoded text of a plain or gzip file, a buffer at a time.  gzip: every member of the file (concatenated members --
// bgzip writes thousands -- decode to the concatenation of their contents, as zlib's own gzread does); bytes behind the
// last member that do not start another one are ignored; a stream that ends early or does not decode is an error.
struct TextSource {
    const std::string &path;
    int fd;
    bool gz;
    IoScratch &sc;
    bool member_open = false, input_eof = false, done = false;
    TextSource(const std::string &p, int f, bool g, IoScratch &s) : path(p), fd(f), gz(g), sc(s)
    {
        if (!gz) return;
        int rc = sc.z_ready ? inflateReset(&sc.strm) : inflateInit2(&sc.strm, 15 + 16);
        if (rc != Z_OK) throw SkError("zlib initialisation failed for " + path);
        sc.z_ready = true;
        sc.strm.avail_in = 0; sc.strm.next_in = sc.zin.data();
        member_open = true;
    }
    void refill()
    {
        // keep what is left (at most a byte or two when looking for the next member's magic) and append
        unsigned char *b = sc.zin.data();
        const size_t left = sc.strm.avail_in;
        if (left && sc.strm.next_in != b) memmove(b, sc.strm.next_in, left);
        const long n = (long)read(fd, b + left, sc.zin.size() - left);
        if (n < 0) throw SkError("read error in " + path);
        if (n == 0) input_eof = true;
        sc.strm.next_in = b; sc.strm.avail_in = (uInt)(left + (size_t)n);
    }
    // up to cap bytes into dst; 0 = end of the text
    long fill(char *dst, size_t cap)
    {
        if (!gz) {
            const long n = (long)read(fd, dst, cap);
            if (n < 0) throw SkError("read error in " + path);
            return n;
        }
        size_t produced = 0;
        while (produced < cap && !done) {
            if (!member_open) {
                // between members: another gzip header, or the end (trailing bytes that are no header are ignored)
                while (sc.strm.avail_in < 2 && !input_eof) refill();
                if (sc.strm.avail_in >= 2 && sc.strm.next_in[0] == 0x1f && sc.strm.next_in[1] == 0x8b) {
                    if (inflateReset(&sc.strm) != Z_OK) throw SkError("zlib reset failed for " + path);
                    member_open = true;
                } else { done = true; break; }
            }
            if (sc.strm.avail_in == 0) {
                if (!input_eof) refill();
                if (sc.strm.avail_in == 0) throw SkError("truncated or corrupt gzip file " + path + ": unexpected end of file");
            }
            sc.strm.next_out = reinterpret_cast<Bytef *>(dst + produced);
            sc.strm.avail_out = (uInt)(cap - produced);
            const int rc = inflate(&sc.strm, Z_NO_FLUSH);
            produced = cap - sc.strm.avail_out;
            if (rc == Z_STREAM_END) member_open = false;
            else if (rc != Z_OK && rc != Z_BUF_ERROR)
                throw SkError("truncated or corrupt gzip file " + path + ": " + (sc.strm.msg ? sc.strm.msg : "zlib data error"));
        }
        return (long)produced;
    }
};
}   // namespace

// the whole text of a plain or gzip file into dst[0 .. cap); SkError("region") if it is longer
static size_t read_text(const std::string &path, bool gz, uint8_t *dst, size_t cap, IoScratch &sc)
{
    FileCloser fc;
    fc.fd = open(path.c_str(), O_RDONLY);
    if (fc.fd < 0) throw SkError("cannot open " + path);
    if (gz) {
        // the compressed file into memory, then decoded straight into the region
        struct stat sb;
        if (fstat(fc.fd, &sb) != 0) throw SkError("cannot open " + path);
        const size_t zn = (size_t)sb.st_size;
        if (sc.zfile.size() < zn) sc.zfile.resize(zn + zn / 4);
        size_t got = 0;
        while (got < zn) {
            const long k = (long)read(fc.fd, sc.zfile.data() + got, zn - got);
            if (k < 0) throw SkError("read error in " + path);
            if (k == 0) break;
            got += (size_t)k;
        }
        size_t n = 0;
        const GunzipStatus st = gunzip_buffer(sc.zfile.data(), got, dst, cap, &n);
        if (st == GUNZIP_OUTPUT_FULL) throw SkError("region");
        if (st != GUNZIP_OK) throw SkError("truncated or corrupt gzip file " + path + ": " + gunzip_status_text(st));
        return n;
    }
    TextSource src(path, fc.fd, gz, sc);
    size_t n = 0;
    for (;;) {
        if (n == cap) {
            char probe;
            if (src.fill(&probe, 1) != 0) throw SkError("region");
            break;
        }
        const long k = src.fill(reinterpret_cast<char *>(dst) + n, cap - n);
        if (k == 0) break;
        n += (size_t)k;
    }
    return n;
}

void read_fasta(const std::string &path, HostGenome &g, uint8_t *region, size_t region_cap, IoScratch *scratch)
{
    g.path = path;
    g.first_name.clear(); g.rec_len.clear(); g.rec_rel.clear(); g.packed_size = 0;
    free(g.own); g.own = nullptr;
    static thread_local IoScratch own_scratch;       // callers without a pool (one-off reads, the sanitizer harness)
    IoScratch &sc = scratch ? *scratch : own_scratch;
    FileCloser fc;
    const bool gz = looks_gzip(path);
    fc.fd = open(path.c_str(), O_RDONLY);
    if (fc.fd < 0) throw SkError("cannot open " + path);
    TextSource src(path, fc.fd, gz, sc);
    OutBuf out{region, 0, region ? region_cap : 0, region != nullptr};
    struct OwnGuard { OutBuf &o; bool armed; ~OwnGuard() { if (armed && !o.fixed) free(o.p); } } guard{out, true};
    std::vector<uint64_t> all_len;
    std::vector<char> &buf = sc.text;
    std::string cur_name;
    bool in_header = false, have_rec = false, have_first = false;
    size_t rec_start = 0;   // offset in the packed layout where the current record starts (multiple of 32)
    // N50 bookkeeping follows util.py:686-724 to the letter: a record's length is the sum of
    // len(line.strip()) over its lines (inner blanks count, empty records are not recorded, text
    // in front of the first header forms a record of its own)
    uint64_t n50_cur = 0, n50_ws = 0;
    bool n50_line_has = false;
    auto close_rec = [&]() {
        if (n50_cur) all_len.push_back(n50_cur);
        n50_cur = 0;
        if (!have_rec) return;
        const size_t len = out.n - rec_start;
        if (len >= ANI_MIN_CONTIG) {
            if (len > 0x7FFFFFFFull) throw SkError("record longer than 2^31 in " + path);
            g.rec_len.push_back((uint32_t)len);
            g.rec_rel.push_back(rec_start);
            if (!have_first) { g.first_name = cur_name; have_first = true; }
            const size_t padded = (len + 31) & ~(size_t)31;
            out.need(padded - len);
            memset(out.p + out.n, 'A', padded - len);
            out.n = rec_start + padded;
        } else {
            out.n = rec_start;   // records below 500 bp are ignored entirely
        }
    };
    bool at_line_start = true;
    for (;;) {
        const long n = src.fill(buf.data(), buf.size());      // (a gzip stream that ends early or does not decode throws: a truncated file
        if (n == 0) break;                                    //  must fail as it does in the reference, not be taken for a shorter genome)
        const char *p = buf.data(), *end = p + n;
        // one pass over the buffer decides between the bulk path (no blank characters: lines are copied with
        // memcpy) and the careful path (CRLF files, blanks inside lines): is any byte below 0x21 not a '\n'?
        bool buf_clean;
        {
            const unsigned char *u = reinterpret_cast<const unsigned char *>(p);
            unsigned bad = 0;
            for (long i = 0; i < n; i++) bad |= (unsigned)(u[i] < 0x21) & (unsigned)(u[i] != '\n');   // vectorised by the compiler
            buf_clean = bad == 0;
        }
        out.need((size_t)n + 64);
        while (p < end) {
            const char *nl = static_cast<const char *>(memchr(p, '\n', (size_t)(end - p)));
            const char *le = nl ? nl : end;                    // end of this piece of the line
            if (in_header) {
                for (const char *q = p; q < le; q++) if (*q != '\r') cur_name.push_back(*q);
                if (nl) { in_header = false; at_line_start = true; }
                p = nl ? nl + 1 : end;
                continue;
            }
            if (at_line_start && p < le && *p == '>') {
                close_rec();
                have_rec = true;
                rec_start = out.n;                             // out.n is a multiple of 32 between records
                cur_name.clear();
                in_header = true;
                at_line_start = false;
                p++;
                continue;
            }
            if (p < le) {
                at_line_start = false;
                if (buf_clean) {
                    const size_t len = (size_t)(le - p);
                    n50_cur += n50_ws + len; n50_ws = 0; n50_line_has = true;
                    if (have_rec) { memcpy(out.p + out.n, p, len); out.n += len; }   // room was made for the whole buffer
                } else {
                    for (const char *q = p; q < le; q++) {
                        const char c = *q;
                        if (c == '\r' || c == ' ' || c == '\t' || c == '\v' || c == '\f') {
                            if (n50_line_has) n50_ws++;
                            continue;
                        }
                        n50_cur += n50_ws + 1; n50_ws = 0; n50_line_has = true;
                        if (have_rec) out.p[out.n++] = (uint8_t)c;
                    }
                }
            }
            if (nl) { at_line_start = true; n50_ws = 0; n50_line_has = false; }
            p = nl ? nl + 1 : end;
        }
    }
    close_rec();
    if (all_len.empty()) throw SkError("no sequence in " + path);
    g.packed_size = out.n;
    if (!out.fixed) { g.own = out.p; guard.armed = false; }
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

// Reader threads of one ingest call: started once, each with its own IoScratch, handed one parallel loop after the other
// (run(n, fn): fn(k, scratch) for k in [0, n), files dealt out one at a time; the first exception is re-thrown).
class IoPool {
  public:
    explicit IoPool(unsigned nthreads)
    {
        for (unsigned t = 0; t < (nthreads ? nthreads : 1u); t++) th_.emplace_back([this]() { worker(); });
    }
    ~IoPool()
    {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
        cv_work_.notify_all();
        for (auto &t : th_) t.join();
    }
    unsigned size() const { return (unsigned)th_.size(); }
    template <class F>
    void run(size_t n, F fn)
    {
        if (n == 0) return;
        std::function<void(size_t, IoScratch &)> f = fn;
        std::unique_lock<std::mutex> lk(mu_);
        fn_ = &f; n_ = n; next_ = 0; pending_ = th_.size(); failed_ = false; first_err_.clear();
        gen_++;
        cv_work_.notify_all();
        cv_done_.wait(lk, [this]() { return pending_ == 0; });
        fn_ = nullptr;
        if (failed_) throw SkError(first_err_);
    }

  private:
    void worker()
    {
        IoScratch sc;
        uint64_t seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> lk(mu_);
            cv_work_.wait(lk, [&]() { return stop_ || gen_ != seen; });
            if (stop_) return;
            seen = gen_;
            for (;;) {
                if (next_ >= n_ || failed_) break;
                const size_t k = next_++;
                lk.unlock();
                try { (*fn_)(k, sc); }
                catch (const std::exception &e) {
                    lk.lock();
                    if (!failed_) { failed_ = true; first_err_ = e.what(); }
                    continue;
                }
                lk.lock();
            }
            if (--pending_ == 0) cv_done_.notify_all();
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_work_, cv_done_;
    const std::function<void(size_t, IoScratch &)> *fn_ = nullptr;
    size_t n_ = 0, next_ = 0, pending_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false, failed_ = false;
    std::string first_err_;
};

namespace {
struct Slot {
    uint8_t *h = nullptr, *d = nullptr; size_t cap = 0, dcap = 0;     // pinned host buffer + device copy (host parse: packed bases; device parse: FASTA text);
                                                                      // the device side may be longer (dcap): texts the device inflates itself live only there
    uint8_t *db = nullptr; size_t db_cap = 0;                  // device parse: the packed bases the kernel writes
    uint32_t *tab = nullptr; size_t tab_cap = 0;               // device parse: pinned landing area of the record tables
    hipStream_t cs = nullptr; hipEvent_t ev = nullptr;         // the slot's copies run on a stream of their own: batch k+1 crosses PCIe while batch k is parsed and sketched
    void release()
    {
        if (d) (void)hipFree(d);
        if (h) (void)hipHostFree(h);
        if (db) (void)hipFree(db);
        if (tab) (void)hipHostFree(tab);
        if (ev) (void)hipEventDestroy(ev);
        if (cs) (void)hipStreamDestroy(cs);
        h = d = db = nullptr; tab = nullptr; cs = nullptr; ev = nullptr; cap = dcap = db_cap = tab_cap = 0;
    }
};
struct StagingSet { Slot sl[2]; bool busy = false; };
std::mutex g_staging_mu;
StagingSet g_staging[64];          // per device
// the device's cached slots if nobody holds them, else a private pair that is freed again
struct StagingLease {
    Slot own[2];
    Slot *slots = own;
    StagingSet *set = nullptr;
    explicit StagingLease(int device)
    {
        if (device < 0 || device >= 64) return;
        std::lock_guard<std::mutex> lk(g_staging_mu);
        if (!g_staging[device].busy) { set = &g_staging[device]; set->busy = true; slots = set->sl; }
    }
    ~StagingLease()
    {
        if (set) { std::lock_guard<std::mutex> lk(g_staging_mu); set->busy = false; return; }
        for (auto &s : own) s.release();
    }
};
}   // namespace

// the cached staging set of a device (pinned text buffer, its device copy, bases buffer, pinned tables: ~0.6 GB of pinned host memory
// and ~1.2 GB of HBM at the default batch size) handed back; a set that is in use stays.  A long-lived host application calls this
// (skder_amd_release_cached_buffers) when it is done with a device for a while.
bool staging_release(int device)
{
    if (device < 0 || device >= 64) return false;
    std::lock_guard<std::mutex> lk(g_staging_mu);
    if (g_staging[device].busy) return false;
    for (auto &sl : g_staging[device].sl) sl.release();
    return true;
}

// Host threads of the ingest: one per file in flight -- as many as the process may actually RUN at once.  A container is
// often given fewer CPUs than the machine shows (cgroup CPU quota: the MI355X boxes this was measured on show 256 hardware
// threads and grant 16 CPUs' worth of time); threads beyond the quota only get throttled, and the ingest was slower with 128
// threads than with 16 there.  SKDER_AMD_IO_THREADS overrides.
static unsigned cgroup_cpu_quota()
{
    // cgroup v2: "<quota> <period>" or "max <period>"; v1: cpu.cfs_quota_us (-1: none) / cpu.cfs_period_us
    double quota = -1, period = 100000;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[64] = {0};
        if (fscanf(f, "%63s %lf", q, &period) == 2 && strcmp(q, "max") != 0) quota = atof(q);
        fclose(f);
    } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
        if (fscanf(g, "%lf", &quota) != 1) quota = -1;
        fclose(g);
        if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(h, "%lf", &period) != 1) period = 100000; fclose(h); }
    }
    if (quota <= 0 || period <= 0) return 0;
    return (unsigned)std::max(1.0, quota / period + 0.5);
}
unsigned ingest_threads()
{
    if (const char *e = getenv("SKDER_AMD_IO_THREADS")) return (unsigned)std::max(1, atoi(e));
    static const unsigned n = []() {
        unsigned t = std::max(1u, std::min(128u, std::thread::hardware_concurrency()));
        const unsigned q = cgroup_cpu_quota();
        return q ? std::min(t, q) : t;
    }();
    return n;
}

// N50 of a list of record lengths as util.py:686-724 computes it (all records; half = int(sum / 2); descending; first cumulative >= half)
static uint64_t n50_of(std::vector<uint64_t> &all_len)
{
    std::sort(all_len.begin(), all_len.end());
    uint64_t tot = 0;
    for (uint64_t l : all_len) tot += l;
    const uint64_t half = tot / 2;
    uint64_t cum = 0, n50 = all_len[0];
    for (size_t i = all_len.size(); i-- > 0;) {
        cum += all_len[i];
        if (cum >= half) { n50 = all_len[i]; break; }
    }
    return n50;
}

// The device stage of a batch whose pinned buffer holds FASTA TEXT (already on its way to d_text on the context's stream): the
// kernels of fasta.hip build the packed layout in d_bases and the record tables; the host adds what only it can (names from the
// header lines, N50 from the length lists) and parses the files the kernel declined (blanks inside sequence lines, ...) itself.
static void device_parse(skder_sketches *s, Slot &S,
                         std::vector<FastaFile> &ff, uint64_t out_total, uint64_t table_total, std::vector<HostGenome> &gs,
                         std::vector<uint64_t> &rec_off, std::vector<uint32_t> &rec_len, std::vector<uint32_t> &gbegin)
{
    hipStream_t st = s->ctx->stream;
    const uint32_t nf = (uint32_t)ff.size();
    const uint8_t *h_text = S.h, *d_text = S.d;
    if (out_total > S.db_cap) {
        if (S.db) (void)hipFree(S.db);
        S.db = nullptr;
        S.db_cap = out_total + out_total / 8;
        HIPCHECK(hipMalloc(&S.db, S.db_cap));
    }
    if (3 * (table_total + 1) > S.tab_cap) {
        if (S.tab) (void)hipHostFree(S.tab);
        S.tab = nullptr;
        S.tab_cap = 3 * (table_total + 1) + (table_total + 1) / 2;
        HIPCHECK(hipHostMalloc(&S.tab, S.tab_cap * sizeof(uint32_t)));
    }
    uint8_t *d_bases = S.db;
    DevBuf<FastaFile> d_ff;
    DevBuf<FastaResult> d_res;
    DevBuf<uint32_t> d_rel, d_len, d_all;
    d_ff.resize(nf, st); d_res.resize(nf, st);
    d_rel.resize(table_total + 1, st); d_len.resize(table_total + 1, st); d_all.resize(table_total + 1, st);
    HIPCHECK(hipMemsetAsync(d_bases, 'A', 32, st));
    // tiles of the files (a file without text has none), then the parser: tiled (three kernels, a wavefront per 4 KB of text), or
    // one wavefront per file (SKDER_AMD_FASTA_WAVE=1: round 3's first device parser; same results)
    static const bool wave_parser = getenv("SKDER_AMD_FASTA_WAVE") != nullptr;
    DevBuf<uint8_t> d_work;
    if (wave_parser) {
        HIPCHECK(hipMemcpyAsync(d_ff.p, ff.data(), nf * sizeof(FastaFile), hipMemcpyHostToDevice, st));
        fasta_parse_launch(d_text, d_ff.p, nf, d_bases, d_rel.p, d_len.p, d_all.p, d_res.p, st);
    } else {
        uint64_t tiles = 0;
        for (uint32_t k = 0; k < nf; k++) { ff[k].tile_off = (uint32_t)tiles; tiles += ((uint64_t)ff[k].text_len + 4095u) / 4096u; }
        if (tiles > 0xFFFFFFF0ull) throw SkError("internal error: too many tiles in one ingest batch");
        HIPCHECK(hipMemcpyAsync(d_ff.p, ff.data(), nf * sizeof(FastaFile), hipMemcpyHostToDevice, st));
        d_work.resize(fasta_tiles_work_bytes((uint32_t)tiles, nf), st);
        fasta_parse_tiles_launch(d_text, d_ff.p, nf, (uint32_t)tiles, d_work.p, d_bases, d_rel.p, d_len.p, d_all.p, d_res.p, st);
    }
    std::vector<FastaResult> res(nf);
    uint32_t *rel = S.tab, *len = S.tab + (table_total + 1), *all = S.tab + 2 * (table_total + 1);
    HIPCHECK(hipMemcpyAsync(res.data(), d_res.p, nf * sizeof(FastaResult), hipMemcpyDeviceToHost, st));
    HIPCHECK(hipMemcpyAsync(rel, d_rel.p, table_total * 4, hipMemcpyDeviceToHost, st));
    HIPCHECK(hipMemcpyAsync(len, d_len.p, table_total * 4, hipMemcpyDeviceToHost, st));
    HIPCHECK(hipMemcpyAsync(all, d_all.p, table_total * 4, hipMemcpyDeviceToHost, st));
    HIPCHECK(hipStreamSynchronize(st));
    rec_off.clear(); rec_len.clear(); gbegin.clear();
    const char *dbg_env = getenv("SKDER_AMD_DEBUG");
    const bool dbg_fasta = dbg_env && atoi(dbg_env) >= 2;          // SKDER_AMD_DEBUG=2: one line per file with what the parser found
    for (uint32_t k = 0; k < nf; k++) {
        const FastaFile &f = ff[k];
        const FastaResult &r = res[k];
        HostGenome &g = gs[k];
        gbegin.push_back((uint32_t)rec_len.size());
        if (r.flags) {
            // the kernel declined: the host's reader, its layout copied into the file's region
            HostGenome hg;
            read_fasta(g.path, hg);
            if (hg.packed_size > f.out_cap) throw SkError("internal error: host layout of " + g.path + " exceeds its region");
            if (hg.packed_size) HIPCHECK(hipMemcpyAsync(d_bases + f.out_off, hg.own, hg.packed_size, hipMemcpyHostToDevice, st));
            HIPCHECK(hipStreamSynchronize(st));
            for (size_t q = 0; q < hg.rec_len.size(); q++) { rec_off.push_back(f.out_off + hg.rec_rel[q]); rec_len.push_back(hg.rec_len[q]); }
            g.first_name = hg.first_name; g.n50 = hg.n50; g.packed_size = hg.packed_size;
            if (getenv("SKDER_AMD_DEBUG")) fprintf(stderr, "[skder_amd] %s: parsed on the host (device parser flags %u)\n", g.path.c_str(), r.flags);
            continue;
        }
        if (r.n_lens == 0) throw SkError("no sequence in " + g.path);
        if (dbg_fasta)
            fprintf(stderr, "[skder_amd] %s: text %u bytes, %u kept records, %u lengths, first header at %u, packed %u\n", g.path.c_str(), f.text_len, r.n_kept, r.n_lens, r.first_hdr, r.packed_size);
        for (uint32_t q = 0; q < r.n_kept; q++) { rec_off.push_back(f.out_off + rel[f.table_off + q]); rec_len.push_back(len[f.table_off + q]); }
        std::vector<uint64_t> al(all + f.table_off, all + f.table_off + r.n_lens);
        g.n50 = n50_of(al);
        g.first_name.clear();
        if (r.n_kept) {
            const uint8_t *p = h_text + f.text_off + r.first_hdr + 1, *e = h_text + f.text_off + f.text_len;
            for (; p < e && *p != '\n'; p++) if (*p != '\r') g.first_name.push_back((char)*p);
        }
        g.packed_size = r.packed_size;
    }
    gbegin.push_back((uint32_t)rec_len.size());
    // readable 'A's behind the last record (a tile's reads run past its record)
    HIPCHECK(hipMemsetAsync(d_bases + (out_total - (SKDER_TILE + 64)), 'A', SKDER_TILE + 64, st));
}

void sketch_files(skder_sketches *s, const std::vector<std::string> &paths, GenomeNames &names, unsigned threads)
{
    skder_ctx *ctx = s->ctx;
    hipStream_t st = ctx->stream;
    size_t batch_bytes = 256ull << 20;   // per batch: enough files for every host thread, small enough that pinning the two staging buffers stays cheap (1,024 files, first call of a process: 131 / 137 / 139 / 173 ms with 64 / 128 / 256 / 512 MB, later calls 111 / 103 / 95 / 89: profiles/run/r3_ingest_batch.py)
    if (const char *e = getenv("SKDER_AMD_IO_BATCH_MB")) batch_bytes = (size_t)std::max(1, atoi(e)) << 20;
    const bool dbg = getenv("SKDER_AMD_DEBUG") != nullptr;
    auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now();
    // two staging slots (pinned host + device buffer each), kept across batches and grown when a batch needs
    // more: while the device copies and sketches the batch of one slot, the host threads parse the next batch
    // into the other.  The slots outlive the call (one set per device, handed to one caller at a time): pinning
    // a few hundred MB costs tens of milliseconds, which a process that ingests more than once -- a search per external
    // query, a dist after a triangle -- pays once.
    StagingLease lease(ctx->device);
    Slot *slot = lease.slots;
    IoPool pool(std::min<size_t>(threads ? threads : ingest_threads(), paths.size() ? paths.size() : 1));
    // file sizes and kinds up front (parallel stat + magic): a plain file's packed layout is bounded by its size (+ 32 bytes
    // of padding per kept record of >= 500 bases), so it is parsed straight into its region of the pinned staging buffer.  A
    // gzip file says how long its text is in its last four bytes (ISIZE) -- reliably so when it holds ONE member: it gets a
    // region from that and is inflated + parsed straight into it as well (no memory of its own, no second copy).  Files whose
    // ISIZE cannot be the whole text (several members, e.g. bgzip; more than 4 GB) and batches in which a region turns out too
    // small take the two-phase path: into memory of their own first, regions from the exact sizes, then copied in.
    struct FileInfo { uint64_t size = 0, text = 0; bool gz = false, trusted = true; };
    std::vector<FileInfo> info(paths.size());
    pool.run(paths.size(), [&](size_t k, IoScratch &) {
        FileCloser fc;
        fc.fd = open(paths[k].c_str(), O_RDONLY);
        struct stat sb;
        if (fc.fd < 0 || fstat(fc.fd, &sb) != 0) throw SkError("cannot open " + paths[k]);
        FileInfo &fi = info[k];
        fi.size = fi.text = (uint64_t)sb.st_size;
        unsigned char m[2] = {0, 0}, tail[4] = {0, 0, 0, 0};
        fi.gz = fi.size >= 2 && pread(fc.fd, m, 2, 0) == 2 && m[0] == 0x1f && m[1] == 0x8b;
        if (fi.gz) {
            uint64_t isize = 0;
            if (fi.size >= 18 && pread(fc.fd, tail, 4, (off_t)fi.size - 4) == 4)
                isize = (uint64_t)tail[0] | (uint64_t)tail[1] << 8 | (uint64_t)tail[2] << 16 | (uint64_t)tail[3] << 24;
            // FASTA text deflates 3-4.5 x: a text shorter than the file, or more than 64 x longer, is not one member's size
            fi.trusted = isize >= fi.size && isize <= 64 * fi.size;
            fi.text = fi.trusted ? isize : 4 * fi.size;
        }
    });
    auto bound = [](uint64_t fsize) { return ((fsize + fsize / 15 + 256) + 31) & ~(uint64_t)31; };
    // the batch starting at i0: files until ~batch_bytes of layout
    auto batch_end = [&](size_t i0) {
        size_t i1 = i0;
        uint64_t est = 0;
        const uint64_t budget = i0 == 0 ? batch_bytes / 4 : batch_bytes;      // a short first batch: the device starts early
        while (i1 < paths.size() && (i1 == i0 || est < budget)) {
            est += bound(info[i1].text);
            i1++;
        }
        return i1;
    };
    struct Prepared {
        size_t i0 = 0, i1 = 0;
        int sl = 0;
        std::vector<HostGenome> gs;
        std::vector<uint64_t> rec_off;
        std::vector<uint32_t> rec_len, gbegin;
        uint64_t total = 0;
        double ms = 0;
        // device parse: the slot's pinned buffer holds FASTA text; the kernel of fasta.hip builds the layout
        bool dev_parse = false;
        std::vector<FastaFile> ff;
        uint64_t out_total = 0, table_total = 0;
    };
    const bool want_dev_parse = getenv("SKDER_AMD_HOST_PARSE") == nullptr;
    auto prepare_host = [&](size_t i0, size_t i1, int sl) {
        const double t0 = now();
        HIPCHECK(hipSetDevice(ctx->device));          // may run on a helper thread
        Prepared P;
        P.i0 = i0; P.i1 = i1; P.sl = sl;
        const size_t ng = i1 - i0;
        std::vector<HostGenome> &gs = P.gs;
        std::vector<uint64_t> gbase(ng + 1);
        uint64_t off = 32;
        Slot &S = slot[sl];
        auto room = [&](uint64_t total) {            // regions: 32 readable bytes in front, SKDER_TILE + 64 behind
            uint64_t device_total = total;
            if (total > S.cap) {
                if (i1 - i0 < paths.size()) total = std::max<uint64_t>(total, batch_bytes + batch_bytes / 16);     // several batches: full size at once (the first batch is a short one)
                if (S.h) (void)hipHostFree(S.h);
                S.h = nullptr;
                S.cap = total + total / 8;
                HIPCHECK(hipHostMalloc(&S.h, S.cap));
            }
            if (device_total > S.dcap) {
                if (device_total < S.cap) device_total = S.cap;
                if (S.d) (void)hipFree(S.d);
                S.d = nullptr;
                S.dcap = device_total + device_total / 8;
                HIPCHECK(hipMalloc(&S.d, S.dcap));
            }
        };
        bool direct = getenv("SKDER_AMD_IO_TWO_PHASE") == nullptr;
        for (size_t k = 0; k < ng; k++) direct = direct && info[i0 + k].trusted;
        bool text_only = direct && want_dev_parse;
        for (size_t k = 0; k < ng && text_only; k++) text_only = bound(info[i0 + k].text) <= 0xFFFFFFFFull;       // (its output region is addressed in 32 bits)
        if (text_only) {
            // DEVICE PARSE: the host threads only read (or inflate) every file into the pinned buffer -- '\n' in front of each
            // text and 128 bytes of '\n' behind it --; fasta_parse_kernel does the scan on the device
            gs.clear(); gs.resize(ng);
            P.ff.assign(ng, FastaFile());
            uint64_t toff = 64, ooff = 32, tabs = 0;
            for (size_t k = 0; k < ng; k++) {
                FastaFile &f = P.ff[k];
                f.text_len = (uint32_t)info[i0 + k].text;      // (gzip: the trailer's figure; the true length follows)
                f.out_off = ooff; f.out_cap = (uint32_t)bound(info[i0 + k].text);
                ooff += f.out_cap;
                f.rec_cap = (uint32_t)(info[i0 + k].text / 256u + 64u); f.table_off = (uint32_t)tabs;
                tabs += f.rec_cap;
                f.text_off = toff;
                toff += (info[i0 + k].text + 128u + 63u) & ~(uint64_t)63u;
            }
            room(toff + 64);
            std::atomic<bool> too_small(false);
            pool.run(ng, [&](size_t k, IoScratch &sc) {
                if (too_small.load()) return;
                FastaFile &f = P.ff[k];
                uint8_t *t = S.h + f.text_off;
                try {
                    const size_t n = read_text(paths[i0 + k], info[i0 + k].gz, t, info[i0 + k].text, sc);
                    f.text_len = (uint32_t)n;
                    memset(t - 64, '\n', 64);
                    memset(t + n, '\n', (((size_t)info[i0 + k].text + 128u + 63u) & ~(size_t)63u) - n);
                    gs[k].path = paths[i0 + k];
                } catch (const SkError &e) {
                    if (std::string(e.what()) != "region") throw;
                    too_small.store(true);              // the text is longer than the file said: the whole batch again, on the host
                }
            });
            if (!too_small.load()) {
                P.dev_parse = true;
                P.total = toff + 64; P.out_total = ooff + SKDER_TILE + 64; P.table_total = tabs;
                P.ms = now() - t0;
                return P;
            }
            direct = false;
        }
        if (direct) {
            // every file straight into its region
            gs.clear(); gs.resize(ng);
            off = 32;
            for (size_t k = 0; k < ng; k++) { gbase[k] = off; off += bound(info[i0 + k].text); }
            gbase[ng] = off;
            room(off + SKDER_TILE + 64);
            std::atomic<bool> too_small(false);
            pool.run(ng, [&](size_t k, IoScratch &sc) {
                if (too_small.load()) return;
                try { read_fasta(paths[i0 + k], gs[k], S.h + gbase[k], gbase[k + 1] - gbase[k], &sc); }
                catch (const SkError &e) {
                    if (std::string(e.what()) != "region") throw;
                    too_small.store(true);              // the text is longer than the file said: the whole batch again, two-phase
                }
            });
            direct = !too_small.load();
        }
        if (!direct) {
            gs.clear(); gs.resize(ng);
            // phase A: gzip files into memory of their own (exact sizes afterwards)
            pool.run(ng, [&](size_t k, IoScratch &sc) { if (info[i0 + k].gz) read_fasta(paths[i0 + k], gs[k], nullptr, 0, &sc); });
            off = 32;
            for (size_t k = 0; k < ng; k++) {
                gbase[k] = off;
                off += info[i0 + k].gz ? gs[k].packed_size : bound(info[i0 + k].size);
            }
            gbase[ng] = off;
            room(off + SKDER_TILE + 64);
            // phase B: plain files are parsed straight into their regions, inflated ones are copied in
            pool.run(ng, [&](size_t k, IoScratch &sc) {
                if (info[i0 + k].gz) {
                    if (gs[k].packed_size) memcpy(S.h + gbase[k], gs[k].own, gs[k].packed_size);
                    free(gs[k].own); gs[k].own = nullptr;
                } else {
                    read_fasta(paths[i0 + k], gs[k], S.h + gbase[k], gbase[k + 1] - gbase[k], &sc);
                }
            });
        }
        P.total = off + SKDER_TILE + 64;
        uint8_t *h = S.h;
        memset(h, 'A', 32);
        memset(h + off, 'A', SKDER_TILE + 64);
        for (size_t k = 0; k < ng; k++) {
            P.gbegin.push_back((uint32_t)P.rec_len.size());
            for (size_t r = 0; r < gs[k].rec_len.size(); r++) {
                P.rec_off.push_back(gbase[k] + gs[k].rec_rel[r]);
                P.rec_len.push_back(gs[k].rec_len[r]);
            }
        }
        P.gbegin.push_back((uint32_t)P.rec_len.size());
        P.ms = now() - t0;
        return P;
    };
    // ... and its copy to the device, on the slot's own stream (the slot's buffers are free: the batch that used them two
    // rounds ago was copied, parsed and sketched before this one was asked for)
    auto prepare = [&](size_t i0, size_t i1, int sl) {
        Prepared P = prepare_host(i0, i1, sl);
        Slot &S = slot[sl];
        if (!S.cs) HIPCHECK(hipStreamCreateWithFlags(&S.cs, hipStreamNonBlocking));
        if (!S.ev) HIPCHECK(hipEventCreateWithFlags(&S.ev, hipEventDisableTiming));
        HIPCHECK(hipMemcpyAsync(S.d, S.h, P.total, hipMemcpyHostToDevice, S.cs));
        HIPCHECK(hipEventRecord(S.ev, S.cs));
        return P;
    };
    double t_parse = 0, t_dev = 0;
    if (paths.empty()) return;
    Prepared cur = prepare(0, batch_end(0), 0);
    for (;;) {
        // the next batch is parsed on a helper thread while this one is copied and sketched
        std::future<Prepared> next;
        const bool more = cur.i1 < paths.size();
        if (more) {
            const size_t n0 = cur.i1, n1 = batch_end(cur.i1);
            const int nsl = cur.sl ^ 1;
            next = std::async(std::launch::async, [&prepare, n0, n1, nsl]() { return prepare(n0, n1, nsl); });
        }
        const double t2 = now();
        t_parse += cur.ms;
        try {
            Slot &S = slot[cur.sl];
            HIPCHECK(hipStreamWaitEvent(st, S.ev, 0));
            if (cur.dev_parse) device_parse(s, S, cur.ff, cur.out_total, cur.table_total, cur.gs, cur.rec_off, cur.rec_len, cur.gbegin);
            skder_batch_t b;
            b.n_genomes = (uint32_t)cur.gs.size();
            b.n_records = (uint32_t)cur.rec_len.size();
            b.rec_off = cur.rec_off.data(); b.rec_len = cur.rec_len.data(); b.genome_rec_begin = cur.gbegin.data();
            sketch_batch_impl(s, cur.dev_parse ? S.db : S.d, &b);
            HIPCHECK(hipStreamSynchronize(st));
        } catch (...) {
            if (more) { try { (void)next.get(); } catch (...) {} }
            for (int k = 0; k < 2; k++) if (slot[k].cs) (void)hipStreamSynchronize(slot[k].cs);      // no copy of a slot in flight when its buffers go back
            throw;
        }
        t_dev += now() - t2;
        for (auto &g : cur.gs) {
            names.path.push_back(g.path); names.first_name.push_back(g.first_name); names.n50.push_back(g.n50);
        }
        if (cur.i0 == 0 && more) {
            // more batches follow: size the seed and marker arrays once, extrapolating from this batch
            const double f = 1.1 * (double)paths.size() / (double)cur.i1;
            (void)skder_amd_sketches_reserve(s, (uint64_t)(s->seed_kmer.n * f) + 4096, (uint64_t)(s->markers.n * f) + 4096);
        }
        if (!more) break;
        cur = next.get();
    }
    if (dbg)
        fprintf(stderr, "[skder_amd] ingest of %zu files: %.1f ms wall (read+parse %.1f ms on helper threads, copy+sketch %.1f ms, overlapped)\n",
                paths.size(), now() - t_begin, t_parse, t_dev);
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

// The same model for the inner maps of the parallel row order below: a key carries a VALUE (the record's index) in one word, and the
// two flat arrays are reused from row to row (a thread orders thousands of rows); growth replays exactly what FxMap does -- the old
// table's entries re-placed in bucket order into one of twice the size.
struct FxValMap {
    std::vector<uint64_t> slot, spare;        // key << 32 | value; ~0 = empty
    uint32_t buckets = 0, items = 0;
    void reset() { buckets = 0; items = 0; }
    static void place(std::vector<uint64_t> &t, uint32_t nb, uint64_t kv)
    {
        const uint64_t h = (kv >> 32) * 0x517cc1b727220a95ULL;
        const uint32_t mask = nb - 1;
        uint32_t pos = (uint32_t)(h & mask);
        while (t[pos] != ~0ull) pos = (pos + 1) & mask;
        t[pos] = kv;
    }
    void insert(uint32_t k, uint32_t v)
    {
        if (buckets == 0 || items + 1 > FxMap::capacity(buckets)) {
            const uint32_t nb = buckets ? buckets * 2 : 4;
            if (spare.size() < nb) spare.resize(nb);
            std::fill(spare.begin(), spare.begin() + nb, ~0ull);
            for (uint32_t b = 0; b < buckets; b++) if (slot[b] != ~0ull) place(spare, nb, slot[b]);
            slot.swap(spare);
            buckets = nb;
        }
        place(slot, buckets, (uint64_t)k << 32 | v);
        items++;
    }
};

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
        const bool bad = ferror(f) != 0;
        if (fclose(f) != 0 || bad) { f = nullptr; remove(tmp.c_str()); throw SkError("write error on " + final_name); }
        f = nullptr;
        if (rename(tmp.c_str(), final_name.c_str()) != 0) { remove(tmp.c_str()); throw SkError("cannot rename to " + final_name); }
    }
    ~TmpFile() { if (f) { fclose(f); remove(tmp.c_str()); } }
};

// fn(k) for k in [0, n) on up to nthreads host threads (k dealt out one at a time); the first exception is re-thrown
template <class F>
static void host_parallel_for(size_t n, unsigned nthreads, F fn)
{
    if (n == 0) return;
    const unsigned nt = (unsigned)std::min<size_t>(nthreads ? nthreads : 1u, n);
    if (nt <= 1) { for (size_t k = 0; k < n; k++) fn(k); return; }
    std::atomic<size_t> next(0);
    std::atomic<bool> failed(false);
    std::mutex mu;
    std::string first_err;
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++)
        th.emplace_back([&]() {
            for (;;) {
                const size_t k = next.fetch_add(1);
                if (k >= n || failed.load()) break;
                try { fn(k); }
                catch (const std::exception &e) {
                    std::lock_guard<std::mutex> lk(mu);
                    if (!failed.exchange(true)) first_err = e.what();
                }
            }
        });
    for (auto &t : th) t.join();
    if (failed) throw SkError(first_err);
}

// the simple statement of skani's triangle row order, kept as the reference tests/host_writer_harness.cpp
// holds the in-place version against: rows by Ref through the outer map's bucket order; inside a row the inner map's,
// filled in ascending Query order; --min-af when writing
std::vector<skder_edge_t> triangle_rows_ordered(const std::vector<skder_edge_t> &edges, double min_af_pct)
{
    std::vector<skder_edge_t> E(edges), rows;
    std::sort(E.begin(), E.end(), [](const skder_edge_t &a, const skder_edge_t &b) {
        return a.ref != b.ref ? a.ref < b.ref : a.query < b.query;
    });
    rows.reserve(E.size());
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
            if (passes_min_af(*it, min_af_pct)) rows.push_back(*it);
        }
    }
    return rows;
}

// The same order established IN PLACE on one thread (what runs when the host has no room for the parallel version's scratch copy, and
// for small tables): at 50,000 genomes the edge list is 10^7-10^8 records of 88 bytes, and the copies the
// simple version takes (sorted copy + output) are GBs.  (1) Ref groups are brought into the outer map's order by one
// in-place pass (cycle-leader permutation over per-group windows, as in an American-flag sort); (2) every group is ordered
// by its own inner map, groups in parallel on the host threads; (3) rows below --min-af are squeezed out in place.
// what this process may still allocate: MemAvailable, and the cgroup's limit minus its use where there is one
static uint64_t host_memory_available()
{
    uint64_t avail = ~0ull;
    if (FILE *f = fopen("/proc/meminfo", "r")) {
        char line[256];
        while (fgets(line, sizeof line, f)) {
            unsigned long long kb;
            if (sscanf(line, "MemAvailable: %llu kB", &kb) == 1) { avail = (uint64_t)kb * 1024u; break; }
        }
        fclose(f);
    }
    unsigned long long mx = 0, cur = 0;
    bool have = false;
    if (FILE *f = fopen("/sys/fs/cgroup/memory.max", "r")) { have = fscanf(f, "%llu", &mx) == 1; fclose(f); }
    if (have) {
        if (FILE *f = fopen("/sys/fs/cgroup/memory.current", "r")) { have = fscanf(f, "%llu", &cur) == 1; fclose(f); } else have = false;
        if (have && mx > cur && mx - cur < avail) avail = mx - cur;
    }
    return avail;
}

// fn(t, lo, hi) over T contiguous pieces of [0, n) on T host threads
template <class F>
static void host_parallel_ranges(size_t n, unsigned T, F fn)
{
    host_parallel_for(T, T, [&](size_t t) { fn((unsigned)t, n * t / T, n * (t + 1) / T); });
}

// fn(lo, hi) over contiguous pieces of [0, n) on the ingest's host threads (one piece per 16,384 items at most): for the plain passes over
// an edge list of 10^7 - 10^8 records that the callers make around the row order (index mapping, AF swaps)
void host_parallel_chunks(size_t n, const std::function<void(size_t, size_t)> &fn)
{
    const unsigned T = (unsigned)std::min<size_t>(ingest_threads(), n / 16384 + 1);
    if (T <= 1) { fn(0, n); return; }
    host_parallel_ranges(n, T, [&](unsigned, size_t lo, size_t hi) { fn(lo, hi); });
}

// Round 6: the row order on ALL host threads.  The in-place version below moves 88-byte records one at a time through a
// cycle-leader pass on one thread (90 ns per record on the MI355X box's host: 9 of the 49 s that the default modes take on 14,000
// genomes of one species); here nothing moves until the order is known: (1) per-thread histograms of Ref over
// contiguous pieces of the list, (2) the outer map's bucket order -> a window per Ref, (3) a stable parallel counting scatter of
// record INDICES into the windows, (4) every window ordered by its inner map -- (Query, index) words sorted, FxValMap replayed on
// reused arrays, --min-af applied -- windows in parallel, (5) the surviving records gathered into a scratch block in parallel
// (first touch on all threads) and copied back.  Needs the index array and room for one more copy of the kept records: taken when
// the host has it (host_memory_available), else the in-place version runs.  Same rows in the same order: tests/host_writer_harness.cpp
// holds both against triangle_rows_ordered under the sanitizers.
// the list the parallel row orders keep between calls (at most 256 MB: a small table's ordered copy is built in it and swapped in)
static std::mutex &rows_spare_mutex() { static std::mutex m; return m; }
static std::vector<skder_edge_t> &rows_spare_list() { static std::vector<skder_edge_t> v; return v; }
void rows_order_release_spare()
{
    std::lock_guard<std::mutex> lk(rows_spare_mutex());
    std::vector<skder_edge_t>().swap(rows_spare_list());
}

// The common part of the parallel row orders: records grouped by a 32-bit GROUP id (Ref of a triangle row, Query of a search table),
// the groups laid out in the order `window_order` gives them, every window ordered (and filtered) by `order_window`, which gets the
// window's record indices, rewrites them in output order and returns how many stay.
template <class GroupOf, class WindowOrder, class OrderWindow>
static bool rows_order_parallel(std::vector<skder_edge_t> &E, unsigned T, size_t small_table_bytes, GroupOf group_of, WindowOrder window_order,
                                OrderWindow order_window)
{
    const size_t n = E.size();
    if (n >= 0xFFFFFFF0ull || T < 2) return false;
    if (host_memory_available() < (uint64_t)n * (sizeof(skder_edge_t) + 8u) + (256ull << 20)) return false;
    std::vector<uint32_t> tmax(T, 0);
    host_parallel_ranges(n, T, [&](unsigned t, size_t lo, size_t hi) {
        uint32_t m = 0;
        for (size_t i = lo; i < hi; i++) { const uint32_t g = group_of(E[i]); m = g > m ? g : m; }
        tmax[t] = m;
    });
    uint32_t max_g = 0;
    for (uint32_t m : tmax) max_g = m > max_g ? m : max_g;
    const size_t G = (size_t)max_g + 1;
    if (G * T > (512ull << 20) / 8) return false;                 // (the per-thread counters would be GBs: not this path's case)
    std::vector<uint64_t> cur(G * T, 0);                          // [t * G + g]: records of group g in piece t, then their first slot
    host_parallel_ranges(n, T, [&](unsigned t, size_t lo, size_t hi) {
        uint64_t *c = cur.data() + (size_t)t * G;
        for (size_t i = lo; i < hi; i++) c[group_of(E[i])]++;
    });
    std::vector<uint64_t> cnt(G, 0);
    for (unsigned t = 0; t < T; t++) for (size_t g = 0; g < G; g++) cnt[g] += cur[(size_t)t * G + g];
    std::vector<uint32_t> groups;                                 // the non-empty groups in output order
    window_order(cnt, groups);
    std::vector<uint64_t> group_begin;
    uint64_t at = 0;
    for (uint32_t g : groups) {
        group_begin.push_back(at);
        uint64_t a = at;
        for (unsigned t = 0; t < T; t++) { uint64_t &c = cur[(size_t)t * G + g]; const uint64_t k = c; c = a; a += k; }
        at += cnt[g];
    }
    group_begin.push_back(at);
    const size_t ngroups = groups.size();
    std::vector<uint32_t> idx(n);
    host_parallel_ranges(n, T, [&](unsigned t, size_t lo, size_t hi) {      // stable: a window holds its records in list order
        uint64_t *c = cur.data() + (size_t)t * G;
        for (size_t i = lo; i < hi; i++) idx[c[group_of(E[i])]++] = (uint32_t)i;
    });
    std::vector<uint64_t> kept(ngroups + 1, 0);
    host_parallel_for(ngroups, T, [&](size_t g) {
        kept[g] = order_window(idx.data() + group_begin[g], (size_t)(group_begin[g + 1] - group_begin[g]));
    });
    uint64_t total = 0;
    for (size_t g = 0; g < ngroups; g++) { const uint64_t k = kept[g]; kept[g] = total; total += k; }
    kept[ngroups] = total;
    // a window of a few records per job would be all overhead, one of 10^6 a straggler: jobs of consecutive windows, ~64 K records each
    std::vector<size_t> job_first{0};
    for (size_t g = 0, in_job = 0; g < ngroups; g++) {
        in_job += (size_t)(kept[g + 1] - kept[g]);
        if (in_job >= 65536 && g + 1 < ngroups) { job_first.push_back(g + 1); in_job = 0; }
    }
    job_first.push_back(ngroups);
    auto gather_into = [&](skder_edge_t *dst) {
        host_parallel_for(job_first.size() - 1, T, [&](size_t j) {
            for (size_t g = job_first[j]; g < job_first[j + 1]; g++) {
                const uint32_t *w = idx.data() + group_begin[g];
                skder_edge_t *o = dst + kept[g];
                const size_t k = (size_t)(kept[g + 1] - kept[g]);
                for (size_t i = 0; i < k; i++) o[i] = E[w[i]];
            }
        });
    };
    if ((size_t)total * sizeof(skder_edge_t) <= small_table_bytes) {
        // SMALL tables (the searches of low_mem_greedy order ~50 MB every batch): gathered into a second list that is then SWAPPED with
        // the caller's -- no copy back --, and the list that comes out of the swap is kept for the next call, so that no call after
        // the first touches fresh memory
        std::mutex &spare_mu = rows_spare_mutex();
        std::vector<skder_edge_t> &spare = rows_spare_list();
        std::vector<skder_edge_t> R;
        {
            std::lock_guard<std::mutex> lk(spare_mu);
            R.swap(spare);
        }
        try {
            if (R.size() < (size_t)total) R.resize((size_t)total);      // (only growth touches memory; the records are overwritten)
        } catch (const std::bad_alloc &) { return false; }             // (nothing has moved yet: the in-place version can still run)
        gather_into(R.data());
        R.resize((size_t)total);
        E.swap(R);
        if (R.capacity() * sizeof(skder_edge_t) <= (256ull << 20)) {
            std::lock_guard<std::mutex> lk(spare_mu);
            if (spare.capacity() < R.capacity()) spare.swap(R);
        }
        return true;
    }
    // BIG tables (a triangle of 10^7 - 10^8 rows): a raw scratch block, first touched by all threads in the gather, copied back in
    // parallel and released
    skder_edge_t *R = static_cast<skder_edge_t *>(malloc((size_t)total * sizeof(skder_edge_t)));
    if (!R) return false;                                           // (nothing has moved yet)
    gather_into(R);
    host_parallel_ranges((size_t)total, T, [&](unsigned, size_t lo, size_t hi) {
        if (hi > lo) memcpy(static_cast<void *>(E.data() + lo), R + lo, (hi - lo) * sizeof(skder_edge_t));
    });
    free(R);
    E.resize((size_t)total);
    return true;
}

bool triangle_rows_order_parallel(std::vector<skder_edge_t> &E, double min_af_pct, unsigned T, size_t small_table_bytes)
{
    return rows_order_parallel(E, T, small_table_bytes, [](const skder_edge_t &e) { return e.ref; },
        [](const std::vector<uint64_t> &cnt, std::vector<uint32_t> &groups) {
            FxMap outer;                                             // distinct Refs enter in ascending order; rows leave in bucket order
            for (size_t r = 0; r < cnt.size(); r++) if (cnt[r]) outer.insert(r);
            for (uint32_t b = 0; b < outer.buckets; b++) if (outer.full[b]) groups.push_back((uint32_t)outer.key[b]);
        },
        [&](uint32_t *w, size_t k) -> size_t {
            static thread_local std::vector<uint64_t> keys;
            static thread_local FxValMap inner;
            keys.resize(k);
            for (size_t j = 0; j < k; j++) keys[j] = (uint64_t)E[w[j]].query << 32 | w[j];
            std::sort(keys.begin(), keys.end());                     // ascending Query (one record per pair)
            inner.reset();
            for (size_t j = 0; j < k; j++) inner.insert((uint32_t)(keys[j] >> 32), (uint32_t)keys[j]);
            size_t out = 0;
            for (uint32_t b = 0; b < inner.buckets; b++) {
                const uint64_t kv = inner.slot[b];
                if (kv == ~0ull) continue;
                const uint32_t i = (uint32_t)kv;
                if (passes_min_af(E[i], min_af_pct)) w[out++] = i;
            }
            return out;
        });
}

// search / dist tables on all host threads: windows = queries in ascending order; inside one, ANI descending (as a float, like rect_before),
// then Ref -- a 64-bit key per record, sorted with its index
bool rect_rows_order_parallel(std::vector<skder_edge_t> &E, double min_af_pct, unsigned T, size_t small_table_bytes)
{
    return rows_order_parallel(E, T, small_table_bytes, [](const skder_edge_t &e) { return e.query; },
        [](const std::vector<uint64_t> &cnt, std::vector<uint32_t> &groups) {
            for (size_t q = 0; q < cnt.size(); q++) if (cnt[q]) groups.push_back((uint32_t)q);
        },
        [&](uint32_t *w, size_t k) -> size_t {
            struct KI { uint64_t key; uint32_t i; };
            static thread_local std::vector<KI> ks;
            ks.clear();
            for (size_t j = 0; j < k; j++) {
                const skder_edge_t &e = E[w[j]];
                if (!passes_min_af(e, min_af_pct)) continue;
                const float a = (float)e.ani;
                uint32_t bits;
                memcpy(&bits, &a, 4);
                // floats order like their bit patterns once the sign is folded in: descending ANI = ascending ~ordered(bits)
                const uint32_t ordered = (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);
                ks.push_back(KI{(uint64_t)(~ordered) << 32 | e.ref, w[j]});
            }
            std::sort(ks.begin(), ks.end(), [](const KI &x, const KI &y) { return x.key < y.key; });
            for (size_t j = 0; j < ks.size(); j++) w[j] = ks[j].i;
            return ks.size();
        });
}

void triangle_rows_order_inplace(std::vector<skder_edge_t> &E, double min_af_pct)
{
    const size_t n = E.size();
    const unsigned T = (unsigned)std::min<size_t>(ingest_threads(), n / 16384 + 1);
    if (n >= 65536 && triangle_rows_order_parallel(E, min_af_pct, T)) return;
    triangle_rows_order_serial(E, min_af_pct);
}

void triangle_rows_order_serial(std::vector<skder_edge_t> &E, double min_af_pct)
{
    const size_t n = E.size();
    if (n == 0) return;
    uint32_t max_ref = 0;
    for (const auto &e : E) max_ref = e.ref > max_ref ? e.ref : max_ref;
    std::vector<uint64_t> cnt((size_t)max_ref + 2, 0);
    for (const auto &e : E) cnt[e.ref]++;
    FxMap outer;                                                 // distinct Refs enter in ascending order
    for (uint32_t r = 0; r <= max_ref; r++) if (cnt[r]) outer.insert(r);
    // window [begin, end) of every Ref in the outer map's bucket order
    std::vector<uint64_t> begin((size_t)max_ref + 2, 0), fill((size_t)max_ref + 2, 0);
    std::vector<uint32_t> group_ref;
    std::vector<uint64_t> group_begin;
    uint64_t at = 0;
    for (uint32_t b = 0; b < outer.buckets; b++) {
        if (!outer.full[b]) continue;
        const uint32_t r = (uint32_t)outer.key[b];
        begin[r] = fill[r] = at;
        group_ref.push_back(r); group_begin.push_back(at);
        at += cnt[r];
    }
    group_begin.push_back(at);
    // in-place distribution: every window is filled from its `fill` cursor by swapping misplaced records to where they belong
    for (size_t g = 0; g < group_ref.size(); g++) {
        const uint32_t r = group_ref[g];
        const uint64_t end = begin[r] + cnt[r];
        while (fill[r] < end) {
            const uint32_t have = E[fill[r]].ref;
            if (have == r) { fill[r]++; continue; }
            std::swap(E[fill[r]], E[fill[have]]);               // the record goes to its own window's cursor
            fill[have]++;
        }
    }
    // inner order, group by group
    host_parallel_for(group_ref.size(), ingest_threads(), [&](size_t g) {
        skder_edge_t *lo = E.data() + group_begin[g], *hi = E.data() + group_begin[g + 1];
        std::sort(lo, hi, [](const skder_edge_t &a, const skder_edge_t &b) { return a.query < b.query; });
        FxMap inner;
        for (skder_edge_t *p = lo; p < hi; p++) inner.insert(p->query);
        std::vector<skder_edge_t> tmp;
        tmp.reserve((size_t)(hi - lo));
        for (uint32_t bb = 0; bb < inner.buckets; bb++) {
            if (!inner.full[bb]) continue;
            const uint32_t q = (uint32_t)inner.key[bb];
            const skder_edge_t *it = std::lower_bound(lo, hi, q, [](const skder_edge_t &e, uint32_t qq) { return e.query < qq; });
            tmp.push_back(*it);
        }
        std::copy(tmp.begin(), tmp.end(), lo);
    });
    E.erase(std::remove_if(E.begin(), E.end(), [&](const skder_edge_t &e) { return !passes_min_af(e, min_af_pct); }), E.end());
}

// dist / search tables: grouped by query in listing order; references by ANI descending
static bool rect_before(const skder_edge_t &a, const skder_edge_t &b)
{
    if (a.query != b.query) return a.query < b.query;
    float x = (float)a.ani, y = (float)b.ani;
    if (x != y) return x > y;
    return a.ref < b.ref;
}
void rect_rows_order_inplace(std::vector<skder_edge_t> &E, double min_af_pct)
{
    const unsigned T0 = (unsigned)std::min<size_t>(ingest_threads(), E.size() / 16384 + 1);
    if (E.size() >= 65536 && rect_rows_order_parallel(E, min_af_pct, T0)) return;
    rect_rows_order_serial(E, min_af_pct);
}

void rect_rows_order_serial(std::vector<skder_edge_t> &E, double min_af_pct)
{
    E.erase(std::remove_if(E.begin(), E.end(), [&](const skder_edge_t &e) { return !passes_min_af(e, min_af_pct); }), E.end());
    const size_t n = E.size();
    const unsigned T = (unsigned)std::min<size_t>(ingest_threads(), n / 65536 + 1);
    if (T <= 1) { std::sort(E.begin(), E.end(), rect_before); return; }
    // pieces sorted in parallel, then merged pairwise (std::inplace_merge)
    std::vector<size_t> cut(T + 1);
    for (unsigned t = 0; t <= T; t++) cut[t] = n * t / T;
    host_parallel_for(T, T, [&](size_t t) { std::sort(E.begin() + cut[t], E.begin() + cut[t + 1], rect_before); });
    for (unsigned w = 1; w < T; w *= 2) {
        std::vector<std::pair<size_t, size_t>> jobs;
        for (unsigned t = 0; t + w < T; t += 2 * w) jobs.emplace_back(t, std::min(t + 2 * w, T));
        host_parallel_for(jobs.size(), T, [&](size_t j) {
            const unsigned t = (unsigned)jobs[j].first;
            std::inplace_merge(E.begin() + cut[t], E.begin() + cut[t + w], E.begin() + cut[jobs[j].second], rect_before);
        });
    }
}
std::vector<skder_edge_t> rect_rows_ordered(const std::vector<skder_edge_t> &edges, double min_af_pct)
{
    std::vector<skder_edge_t> E(edges);
    rect_rows_order_inplace(E, min_af_pct);
    return E;
}

// one row of the table appended to `o`
// "\t%.2f" of (double)(frac * 100.0f) appended to p: the hundredths come from skder_amd_pct2_cents (select.cpp: integer arithmetic on the
// float's mantissa, ties to even like glibc's printf -- tests/test_selection_native.py holds it against printf), the digits from two
// divisions; three of these per row were two thirds of a row's formatting time through snprintf.  Values a table never holds (negative,
// NaN, beyond 10^7) take snprintf.
static char *put_pct2(char *p, float frac)
{
    *p++ = '\t';
    const float x = frac * 100.0f;
    if (!(x >= 0.0f) || !(x < 1.0e7f)) return p + snprintf(p, 40, "%.2f", (double)x);
    const uint64_t c = (uint64_t)skder_amd_pct2_cents(frac);
    uint64_t ip = c / 100u;
    const unsigned fr = (unsigned)(c % 100u);
    char tmp[24];
    int k = 0;
    do { tmp[k++] = (char)('0' + ip % 10u); ip /= 10u; } while (ip);
    while (k) *p++ = tmp[--k];
    *p++ = '.';
    *p++ = (char)('0' + fr / 10u);
    *p++ = (char)('0' + fr % 10u);
    return p;
}

// one row of the table appended to `o`
static void format_row(std::string &o, const std::string &rf, const std::string &qf, const skder_edge_t &e, const std::string &rn,
                       const std::string &qn)
{
    // skani keeps its results in single precision and prints percentages with two decimals
    char num[160];
    char *p = put_pct2(num, (float)e.ani);
    p = put_pct2(p, (float)e.af_ref);
    p = put_pct2(p, (float)e.af_query);
    *p++ = '\t';
    o.append(rf); o.push_back('\t'); o.append(qf);
    o.append(num, (size_t)(p - num));
    o.append(rn); o.push_back('\t'); o.append(qn); o.push_back('\n');
}

// Rows are formatted in blocks on the host threads and every block is written at ITS offset of the one output file
// (pwrite): the text of block b starts where block b - 1 ends, which its writer publishes as soon as it knows its own start
// and length -- formatting runs in parallel, only that hand-over is ordered.  (At 50,000 genomes the table is GBs of text.)
void write_rows_tsv(const std::string &out, const skder_edge_t *rows, size_t n, const GenomeNames &ref_names,
                    const GenomeNames &query_names)
{
    TmpFile tf(out);
    const size_t hdr = strlen(TSV_HEADER);
    if (fwrite(TSV_HEADER, 1, hdr, tf.f) != hdr || fflush(tf.f) != 0) throw SkError("write error on " + out);
    const int fd = fileno(tf.f);
    const size_t BLOCK = 16384;
    const size_t nblocks = (n + BLOCK - 1) / BLOCK;
    std::vector<uint64_t> start(nblocks + 1, 0);
    std::vector<char> known(nblocks + 1, 0);
    std::mutex mu;
    std::condition_variable cv;
    bool aborted = false;          // a block failed: nobody waits for an offset that will never be published (host_parallel_for
                                   // stops dealing out blocks after a failure, so the block in front of a waiter may never run)
    auto abort_all = [&]() { { std::lock_guard<std::mutex> lk(mu); aborted = true; } cv.notify_all(); };
    start[0] = hdr; known[0] = 1;
    host_parallel_for(nblocks, ingest_threads(), [&](size_t b) {
        std::string text, failure;
        const size_t lo = b * BLOCK, hi = std::min(n, lo + BLOCK);
        try {
            size_t need = 0;                          // the block's text in ONE allocation: paths and names are ~80 characters each, four per row
            for (size_t i = lo; i < hi; i++) {
                const skder_edge_t &e = rows[i];
                need += ref_names.path[e.ref].size() + query_names.path[e.query].size() + ref_names.first_name[e.ref].size() +
                        query_names.first_name[e.query].size() + 32;
            }
            text.reserve(need);
            for (size_t i = lo; i < hi; i++) {
                const skder_edge_t &e = rows[i];
                format_row(text, ref_names.path[e.ref], query_names.path[e.query], e, ref_names.first_name[e.ref], query_names.first_name[e.query]);
            }
        } catch (const std::exception &e) { failure = e.what()[0] ? e.what() : "out of memory"; text.clear(); }
        uint64_t at;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&]() { return known[b] != 0 || aborted; });          // blocks are dealt out in ascending order: b - 1 is being written or done
            if (!known[b]) throw SkError("edge table: another block failed");
            at = start[b];
            start[b + 1] = at + text.size(); known[b + 1] = 1;     // published whatever happens to this block: nobody waits for ever
        }
        cv.notify_all();
        if (!failure.empty()) { abort_all(); throw SkError("edge table: " + failure); }
        size_t done = 0;
        while (done < text.size()) {
            const ssize_t w = pwrite(fd, text.data() + done, text.size() - done, (off_t)(at + done));
            if (w <= 0) { abort_all(); throw SkError("write error on " + out); }
            done += (size_t)w;
        }
    });
    tf.commit();
}

void write_triangle_tsv(const std::string &out, const std::vector<skder_edge_t> &edges, const GenomeNames &names, double min_af_pct)
{
    std::vector<skder_edge_t> rows(edges);
    triangle_rows_order_inplace(rows, min_af_pct);
    write_rows_tsv(out, rows.data(), rows.size(), names, names);
}

void write_rect_tsv(const std::string &out, const std::vector<skder_edge_t> &edges, const GenomeNames &ref_names,
                    const GenomeNames &query_names, double min_af_pct)
{
    std::vector<skder_edge_t> rows(edges);
    rect_rows_order_inplace(rows, min_af_pct);
    write_rows_tsv(out, rows.data(), rows.size(), ref_names, query_names);
}

void write_n50_tsv(const std::string &out, const GenomeNames &names)
{
    TmpFile tf(out);
    for (size_t i = 0; i < names.path.size(); i++)
        fprintf(tf.f, "%s\t%llu\n", names.path[i].c_str(), (unsigned long long)names.n50[i]);
    tf.commit();
}

// ---------------------------------------------------------------------------------------------
// sketch store.  Layout, little endian:
//   "SKDRAMD1" | u32 version | u32 k, c, marker k, marker c, min contig | u64 n_genomes, n_seeds,
//   n_markers, n_rec_goff | seed_off[n+1] marker_off[n+1] genome_len[n] n50[n] (u64) |
//   genome_nrec[n] rec_goff[n_rec_goff] (u32) | per genome: u32 len + path, u32 len + first name |
//   markers (u64 x n_markers) | seed_kmer, seed_gpos, seed_ctg (u32 x n_seeds) | u64 FNV-1a of all before

static const char STORE_MAGIC[8] = {'S', 'K', 'D', 'R', 'A', 'M', 'D', '1'};

struct StoreIO {
    FILE *f;
    uint64_t h = 0xcbf29ce484222325ULL;
    void mix(const void *p, size_t n)
    {
        // FNV-1a over 8-byte words (tail bytes one by one): cheap enough for multi-GB stores
        const uint8_t *b = (const uint8_t *)p;
        size_t i = 0;
        for (; i + 8 <= n; i += 8) { uint64_t w; memcpy(&w, b + i, 8); h = (h ^ w) * 0x100000001b3ULL; }
        for (; i < n; i++) h = (h ^ b[i]) * 0x100000001b3ULL;
    }
    void put(const void *p, size_t n)
    {
        if (n && fwrite(p, 1, n, f) != n) throw SkError("sketch store: write error");
        mix(p, n);
    }
    void get(void *p, size_t n)
    {
        if (n && fread(p, 1, n, f) != n) throw SkError("sketch store: file is truncated");
        mix(p, n);
    }
    template <class T> void putv(const std::vector<T> &v) { put(v.data(), v.size() * sizeof(T)); }
    template <class T> void getv(std::vector<T> &v, size_t n) { v.resize(n); get(v.data(), n * sizeof(T)); }
    void puts_(const std::string &s) { uint32_t l = (uint32_t)s.size(); put(&l, 4); put(s.data(), l); }
    std::string gets_()
    {
        uint32_t l; get(&l, 4);
        if (l > (1u << 20)) throw SkError("sketch store: corrupt string length");
        std::string s(l, 0); get(&s[0], l);
        return s;
    }
};

void store_save(const std::string &out, skder_sketches *s, const GenomeNames &names)
{
    hipStream_t st = s->ctx->stream;
    const uint64_t n = s->n_genomes, ns = s->h_seed_off.back(), nm = s->h_marker_off.back(), nr = s->h_rec_goff.size();
    if (names.path.size() != n || names.n50.size() != n) throw SkError("sketch store: name table does not match the sketches");
    std::vector<uint32_t> kmer(ns), gpos(ns), ctg(ns);
    std::vector<uint64_t> markers(nm);
    if (ns) {
        HIPCHECK(hipMemcpyAsync(kmer.data(), s->seed_kmer.p, ns * 4, hipMemcpyDeviceToHost, st));
        HIPCHECK(hipMemcpyAsync(gpos.data(), s->seed_gpos.p, ns * 4, hipMemcpyDeviceToHost, st));
        HIPCHECK(hipMemcpyAsync(ctg.data(), s->seed_ctg.p, ns * 4, hipMemcpyDeviceToHost, st));
    }
    if (nm) HIPCHECK(hipMemcpyAsync(markers.data(), s->markers.p, nm * 8, hipMemcpyDeviceToHost, st));
    HIPCHECK(hipStreamSynchronize(st));
    TmpFile tf(out);
    StoreIO io{tf.f};
    io.put(STORE_MAGIC, 8);
    const uint32_t head[6] = {1, ANI_K, ANI_C, ANI_MARKER_K, ANI_MARKER_C, ANI_MIN_CONTIG};
    io.put(head, sizeof head);
    const uint64_t cnt[4] = {n, ns, nm, nr};
    io.put(cnt, sizeof cnt);
    io.putv(s->h_seed_off); io.putv(s->h_marker_off); io.putv(s->h_genome_len); io.putv(names.n50);
    io.putv(s->h_genome_nrec); io.putv(s->h_rec_goff);
    for (uint64_t g = 0; g < n; g++) { io.puts_(names.path[g]); io.puts_(names.first_name[g]); }
    io.putv(markers); io.putv(kmer); io.putv(gpos); io.putv(ctg);
    const uint64_t sum = io.h;
    if (fwrite(&sum, 1, 8, tf.f) != 8) throw SkError("sketch store: write error");
    tf.commit();
}

void store_load(const std::string &path, skder_sketches *s, GenomeNames &names)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) throw SkError("cannot open sketch store " + path);
    struct Closer { FILE *f; ~Closer() { fclose(f); } } closer{f};
    StoreIO io{f};
    char magic[8];
    io.get(magic, 8);
    if (memcmp(magic, STORE_MAGIC, 8) != 0) throw SkError(path + " is not a libskder_amd sketch store");
    uint32_t head[6];
    io.get(head, sizeof head);
    if (head[0] != 1) throw SkError("sketch store: unsupported version " + std::to_string(head[0]));
    if (head[1] != ANI_K || head[2] != ANI_C || head[3] != ANI_MARKER_K || head[4] != ANI_MARKER_C || head[5] != ANI_MIN_CONTIG)
        throw SkError("sketch store was written with different sketching parameters");
    uint64_t cnt[4];
    io.get(cnt, sizeof cnt);
    const uint64_t n = cnt[0], ns = cnt[1], nm = cnt[2], nr = cnt[3];
    if (fseek(f, 0, SEEK_END) != 0) throw SkError("sketch store: cannot seek");
    const uint64_t fsize = (uint64_t)ftell(f);
    const uint64_t fixed = 8 + sizeof head + sizeof cnt + (2 * (n + 1) + 2 * n) * 8 + (n + nr) * 4 + ns * 12 + nm * 8 + 8;
    if (n > 0xFFFFFFFFull || fsize < fixed || nr < n) throw SkError("sketch store: header does not match the file size");
    if (fseek(f, 8 + sizeof head + sizeof cnt, SEEK_SET) != 0) throw SkError("sketch store: cannot seek");
    std::vector<uint64_t> seed_off, marker_off, genome_len;
    std::vector<uint32_t> nrec, rec_goff;
    io.getv(seed_off, n + 1); io.getv(marker_off, n + 1); io.getv(genome_len, n); io.getv(names.n50, n);
    io.getv(nrec, n); io.getv(rec_goff, nr);
    if (seed_off[0] != 0 || seed_off[n] != ns || marker_off[0] != 0 || marker_off[n] != nm)
        throw SkError("sketch store: offset tables are inconsistent");
    uint64_t rsum = 0;
    for (uint64_t g = 0; g < n; g++) {
        if (seed_off[g + 1] < seed_off[g] || marker_off[g + 1] < marker_off[g]) throw SkError("sketch store: offset tables are inconsistent");
        rsum += (uint64_t)nrec[g] + 1;
    }
    if (rsum != nr) throw SkError("sketch store: record tables are inconsistent");
    names.path.resize(n); names.first_name.resize(n);
    for (uint64_t g = 0; g < n; g++) { names.path[g] = io.gets_(); names.first_name[g] = io.gets_(); }
    // payload through pinned memory straight into HBM
    uint8_t *h = nullptr;
    const uint64_t bytes = ns * 12 + nm * 8;
    HIPCHECK(hipHostMalloc(&h, bytes ? bytes : 8));
    struct HostFree { uint8_t *p; ~HostFree() { (void)hipHostFree(p); } } hf{h};
    // the checksum runs over the four arrays one after the other, each with its own word / tail-byte split, as store_save wrote
    // them: one get() over the whole payload would split differently whenever an array's size is not a multiple of 8 (odd
    // number of seeds) and refuse a good file
    io.get(h, nm * 8);
    io.get(h + nm * 8, ns * 4);
    io.get(h + nm * 8 + ns * 4, ns * 4);
    io.get(h + nm * 8 + ns * 8, ns * 4);
    uint64_t sum_file = 0, sum_calc = io.h;
    if (fread(&sum_file, 1, 8, f) != 8) throw SkError("sketch store: file is truncated");
    if (sum_file != sum_calc) throw SkError("sketch store: checksum mismatch (file is corrupt)");
    hipStream_t st = s->ctx->stream;
    DevBuf<uint8_t> d;
    d.resize(bytes ? bytes : 8, st);
    HIPCHECK(hipMemcpyAsync(d.p, h, bytes, hipMemcpyHostToDevice, st));
    HIPCHECK(hipStreamSynchronize(st));
    skder_raw_view_t v;
    v.n_genomes = (uint32_t)n; v.n_seeds = ns; v.n_markers = nm; v.n_rec_goff = nr;
    v.d_markers = (const uint64_t *)d.p;
    v.d_seed_kmer = (const uint32_t *)(d.p + nm * 8); v.d_seed_gpos = v.d_seed_kmer + ns; v.d_seed_ctg = v.d_seed_gpos + ns;
    v.h_seed_off = seed_off.data(); v.h_marker_off = marker_off.data(); v.h_genome_len = genome_len.data();
    v.h_genome_nrec = nrec.data(); v.h_rec_goff = rec_goff.data();
    if (skder_amd_sketches_append_raw(s, &v) != 0) throw SkError(s->ctx->last_error);
}
