// select.cpp -- representative selection on the engine's edge records IN MEMORY; host code only.
//
// Native counterparts of the reference's consumers of the skani edge table, fed from skder_edge_t rows (the array
// skder_amd_db_triangle hands over) instead of the text table, with the reference's exact text conventions so that every
// output file can be compared byte for byte:
//   greedy    src/skDER/skDERsum.cpp:60-165 (connectivity x N50 score, member lists), `sort -k 2 -gr` (skder.py:145-147),
//             the greedy loop (skder.py:150-165)
//   dynamic   src/skDER/skDERcore.cpp:60-224 (two passes; the CODE's rule: af_query - af_subject <= max difference ->
//             the genome with the larger AF is redundant, ties the subject; else the lower N50 x connectivity, ties the subject)
//   clusters  src/skDER/skder.py:168-277 (determineClusters), both branches: with a name mapping (mge_proc_to_unproc_mapping)
//             the names written are the mapped ones
// Both C++ programs of the reference `stod` the two-decimal text of the table: an edge's three values are therefore
// rounded HERE exactly as `%.2f` prints the single-precision percentage (pct2_cents: integer arithmetic on the float's
// mantissa, ties to even like glibc's printf) and compared as the doubles cents / 100.0, which is what stod returns.
// skder_amd/selection.py is the readable statement of the same rules; tests hold the two and the reference's own binaries
// (oracle/_ref, built from /root/reference) against each other.
#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>
#include <unistd.h>

#include "../../include/skder_amd.h"

namespace {

struct SelError : std::runtime_error { using std::runtime_error::runtime_error; };

// `%.2f` of (double)(x * 100.0f), as an integer number of hundredths: the float's exact value m * 2^e times 100, rounded to
// nearest, ties to even (glibc rounds the exact binary value).  x >= 0.
inline int64_t pct2_cents(float frac)
{
    const float x = frac * 100.0f;
    if (!(x > 0.0f)) return 0;
    uint32_t bits;
    memcpy(&bits, &x, 4);
    const int ex = (int)((bits >> 23) & 0xFF);
    uint64_t m = bits & 0x7FFFFFu;
    int e;
    if (ex == 0) e = -149; else { m |= 0x800000u; e = ex - 150; }      // x = m * 2^e
    const uint64_t v = m * 100u;                                        // < 2^31: exact
    if (e >= 0) return (int64_t)(v << e);                               // (percentages stay far below 2^63)
    const int sh = -e;
    if (sh >= 63) return 0;
    const uint64_t q = v >> sh, rem = v & ((1ull << sh) - 1ull), half = 1ull << (sh - 1);
    return (int64_t)(q + ((rem > half) | ((rem == half) & (q & 1ull))));
}

struct Row { uint32_t q, s; int32_t ani, qaf, saf; };       // column 1, column 2 (indices); hundredths of a percent

unsigned n_threads(size_t work)
{
    unsigned t = std::thread::hardware_concurrency();
    if (const char *e = getenv("SKDER_AMD_IO_THREADS")) t = (unsigned)atoi(e);
    if (t < 1) t = 1;
    if (t > 64) t = 64;
    const size_t by_work = work / 262144 + 1;
    return (unsigned)std::min<size_t>(t, by_work);
}
template <typename F>
void parallel_ranges(size_t n, F fn)
{
    const unsigned T = n_threads(n);
    if (T <= 1) { fn(0u, (size_t)0, n); return; }
    std::vector<std::thread> th;
    for (unsigned t = 1; t < T; t++) th.emplace_back(fn, t, n * t / T, n * (t + 1) / T);
    fn(0u, (size_t)0, n / T);
    for (auto &x : th) x.join();
}

std::vector<Row> rows_of(const skder_edge_t *e, uint64_t n, uint32_t n_genomes)
{
    std::vector<Row> r(n);
    std::atomic<bool> bad(false);
    parallel_ranges(n, [&](unsigned, size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; i++) {
            if (e[i].ref >= n_genomes || e[i].query >= n_genomes) { bad = true; return; }
            r[i].q = e[i].ref; r[i].s = e[i].query;
            r[i].ani = (int32_t)pct2_cents((float)e[i].ani);
            r[i].qaf = (int32_t)pct2_cents((float)e[i].af_ref);
            r[i].saf = (int32_t)pct2_cents((float)e[i].af_query);
        }
    });
    if (bad) throw SelError("edge record with a genome index beyond n_genomes");
    return r;
}

// the smallest number of hundredths c with c / 100.0 >= cut (the comparison the reference makes on the parsed text)
int32_t cents_at_least(double cut)
{
    if (!(cut > -1e9)) return INT32_MIN;
    if (cut > 1e7) return INT32_MAX;
    int64_t c = (int64_t)std::floor(cut * 100.0) - 2;
    while ((double)c / 100.0 < cut) c++;
    return (int32_t)c;
}

struct TmpOut {
    std::string path, tmp;
    FILE *f = nullptr;
    explicit TmpOut(const char *p)
    {
        if (!p) return;
        path = p; tmp = path + ".tmp." + std::to_string((long)getpid());
        f = fopen(tmp.c_str(), "wb");
        if (!f) throw SelError("cannot write " + path + ": " + strerror(errno));
    }
    void put(const std::string &s) { if (f && fwrite(s.data(), 1, s.size(), f) != s.size()) throw SelError("write error on " + path); }
    void commit()
    {
        if (!f) return;
        const bool ok = fclose(f) == 0;
        f = nullptr;
        if (!ok || rename(tmp.c_str(), path.c_str()) != 0) { remove(tmp.c_str()); throw SelError("write error on " + path); }
    }
    ~TmpOut() { if (f) { fclose(f); remove(tmp.c_str()); } }
};

// C++ `ostream << double` with default precision (skDERsum.cpp:53-59): %g, six significant digits
std::string fmt_score(double x) { char b[64]; snprintf(b, sizeof b, "%g", x); return b; }

// Python's str(float) of a two-decimal value: shortest round trip = the decimals without trailing zeros, at least one
std::string py_float(int32_t cents)
{
    char b[32];
    const int32_t w = cents / 100, d = cents % 100;
    if (d == 0) snprintf(b, sizeof b, "%d.0", w);
    else if (d % 10 == 0) snprintf(b, sizeof b, "%d.%d", w, d / 10);
    else snprintf(b, sizeof b, "%d.%02d", w, d);
    return b;
}

struct Names {
    const char *const *paths, *const *display;
    const char *path(uint32_t g) const { return paths[g]; }
    const char *shown(uint32_t g) const { return display ? display[g] : paths[g]; }
};

void check_args(const void *rows, uint64_t n_rows, uint32_t n_genomes, const char *const *paths)
{
    if ((!rows && n_rows) || !paths) throw SelError("null argument");
    for (uint32_t g = 0; g < n_genomes; g++) if (!paths[g]) throw SelError("null path");
}

// ---- greedy ---------------------------------------------------------------------------------------------------------------
void greedy(const skder_edge_t *edges, uint64_t n_rows, uint32_t n_genomes, const Names &nm, const uint64_t *n50, double min_ani, double min_af,
            const char *info_txt, const char *sorted_txt, const char *results_txt, uint32_t *reps, uint32_t *n_reps)
{
    const std::vector<Row> R = rows_of(edges, n_rows, n_genomes);
    const int32_t c_ani = cents_at_least(min_ani), c_af = cents_at_least(min_af);
    // connectivity and member lists (skDERsum.cpp:112-125): an edge counts if ani >= min and one of the two fractions does; then
    // column 1 gains column 2 as a member when column 5 (the subject's fraction) passes, column 2 gains column 1 when column 4 does.
    // Lists are in table order: counted, offset, filled (CSR)
    std::vector<uint32_t> conn(n_genomes, 0);
    for (const Row &r : R)
        if (r.ani >= c_ani) { if (r.saf >= c_af) conn[r.q]++; if (r.qaf >= c_af) conn[r.s]++; }
    std::vector<uint64_t> off(n_genomes + 1, 0);
    for (uint32_t g = 0; g < n_genomes; g++) off[g + 1] = off[g] + conn[g];
    std::vector<uint32_t> mem(off[n_genomes]);
    {
        std::vector<uint64_t> at(off.begin(), off.end() - 1);
        for (const Row &r : R)
            if (r.ani >= c_ani) { if (r.saf >= c_af) mem[at[r.q]++] = r.s; if (r.qaf >= c_af) mem[at[r.s]++] = r.q; }
    }
    // Genome_Information_for_Greedy_Clustering.txt, N50-file (= listing) order; the sort key is the TEXT of the score as
    // `sort -g` reads it back: two scores that print alike with six digits tie
    std::vector<double> key(n_genomes);
    std::vector<std::string> score_txt(n_genomes);
    for (uint32_t g = 0; g < n_genomes; g++) {
        if (conn[g]) { score_txt[g] = fmt_score((double)(int)n50[g] * (double)conn[g]); key[g] = strtod(score_txt[g].c_str(), nullptr); }
        else { score_txt[g] = "0.0"; key[g] = 0.0; }
    }
    auto line_of = [&](uint32_t g, std::string &o) {
        o.append(nm.path(g)); o.push_back('\t'); o.append(score_txt[g]); o.push_back('\t');
        for (uint64_t k = off[g]; k < off[g + 1]; k++) { if (k > off[g]) o.append("; "); o.append(nm.path(mem[k])); }
        o.push_back('\n');
    };
    auto write_lines = [&](const char *file, const std::vector<uint32_t> *order) {
        if (!file) return;
        TmpOut out(file);
        std::string buf;
        for (uint32_t i = 0; i < n_genomes; i++) {
            line_of(order ? (*order)[i] : i, buf);
            if (buf.size() > (1u << 22)) { out.put(buf); buf.clear(); }
        }
        out.put(buf);
        out.commit();
    };
    write_lines(info_txt, nullptr);
    // `sort -k 2 -gr` in the C locale: general-numeric key descending; ties by the whole line, bytes, reversed too.  Lines differ
    // inside "path<TAB>" (paths are unique and hold no tab), so the last resort compares that
    std::vector<uint32_t> order(n_genomes);
    for (uint32_t g = 0; g < n_genomes; g++) order[g] = g;
    std::vector<std::string> pt(n_genomes);
    for (uint32_t g = 0; g < n_genomes; g++) { pt[g] = nm.path(g); pt[g].push_back('\t'); }
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
        if (key[a] != key[b]) return key[a] > key[b];
        return pt[a].compare(pt[b]) > 0;          // (std::string::compare is memcmp-like on unsigned bytes via char_traits)
    });
    write_lines(sorted_txt, &order);
    // skder.py:150-165: in sorted order, a genome not yet accounted for is a representative and accounts for its members
    std::vector<uint8_t> accounted(n_genomes, 0);
    uint32_t nr = 0;
    TmpOut res(results_txt);
    std::string buf;
    for (uint32_t i = 0; i < n_genomes; i++) {
        const uint32_t g = order[i];
        if (accounted[g]) continue;
        for (uint64_t k = off[g]; k < off[g + 1]; k++) accounted[mem[k]] = 1;
        if (reps) reps[nr] = g;
        nr++;
        buf.append(nm.shown(g)); buf.push_back('\n');
    }
    res.put(buf);
    res.commit();
    if (n_reps) *n_reps = nr;
}

// ---- dynamic --------------------------------------------------------------------------------------------------------------
void dynamic(const skder_edge_t *edges, uint64_t n_rows, uint32_t n_genomes, const Names &nm, const uint64_t *n50, double min_ani, double min_af,
             double max_af_diff, const char *results_txt, uint32_t *reps, uint32_t *n_reps)
{
    const std::vector<Row> R = rows_of(edges, n_rows, n_genomes);
    const int32_t c_ani = cents_at_least(min_ani), c_af = cents_at_least(min_af);
    std::vector<uint32_t> conn(n_genomes, 0);        // skDERcore.cpp:95-98: both ends
    for (const Row &r : R)
        if (r.ani >= c_ani && (r.qaf >= c_af || r.saf >= c_af)) { conn[r.q]++; conn[r.s]++; }
    std::vector<uint8_t> redundant(n_genomes, 0);
    for (const Row &r : R) {
        if (!(r.ani >= c_ani && (r.qaf >= c_af || r.saf >= c_af))) continue;
        const double afq = (double)r.qaf / 100.0, afs = (double)r.saf / 100.0;
        if (afq - afs <= max_af_diff) redundant[afq > afs ? r.q : r.s] = 1;                       // :171-177
        else {
            const double qs = (double)(int)n50[r.q] * (double)(int)conn[r.q], ss = (double)(int)n50[r.s] * (double)(int)conn[r.s];
            redundant[qs >= ss ? r.s : r.q] = 1;                                                  // :178-185
        }
    }
    uint32_t nr = 0;
    TmpOut res(results_txt);
    std::string buf;
    for (uint32_t g = 0; g < n_genomes; g++) {       // survivors in N50-file order (:200-216)
        if (redundant[g]) continue;
        if (reps) reps[nr] = g;
        nr++;
        buf.append(nm.shown(g)); buf.push_back('\n');
    }
    res.put(buf);
    res.commit();
    if (n_reps) *n_reps = nr;
}

// ---- secondary clustering -------------------------------------------------------------------------------------------------
struct Best { int32_t ani = 0, af = 0; std::vector<uint32_t> who; bool touched = false; };

void clusters(const skder_edge_t *edges, uint64_t n_rows, uint32_t n_genomes, const Names &nm, const uint32_t *reps, uint32_t n_reps,
              double af_cutoff, double ani_cutoff, const char *out_txt)
{
    const std::vector<Row> R = rows_of(edges, n_rows, n_genomes);
    const int32_t c_ani = cents_at_least(ani_cutoff), c_af = cents_at_least(af_cutoff);
    std::vector<uint8_t> is_rep(n_genomes, 0);
    for (uint32_t i = 0; i < n_reps; i++) { if (reps[i] >= n_genomes) throw SelError("representative index beyond n_genomes"); is_rep[reps[i]] = 1; }
    // best match per non-representative, in a strict table (its fraction passes the cut-off) and a loose one; the tables keep the
    // order in which genomes were first looked up (a Python dict), the names of equal matches the order in which they came
    std::vector<Best> strict(n_genomes), loose(n_genomes);
    std::vector<uint32_t> strict_order, loose_order;
    auto upd = [&](std::vector<Best> &tab, std::vector<uint32_t> &ord, uint32_t g, uint32_t other, int32_t ani, int32_t af) {
        Best &b = tab[g];
        if (!b.touched) { b.touched = true; ord.push_back(g); }
        if (ani > b.ani || (ani == b.ani && af > b.af)) { b.ani = ani; b.af = af; b.who.assign(1, other); }
        else if (ani == b.ani && af == b.af) { if (std::find(b.who.begin(), b.who.end(), other) == b.who.end()) b.who.push_back(other); }
    };
    for (const Row &r : R) {          // column 1 = ref, column 2 = que, column 4 = raf, column 5 = qaf (skder.py:190-228)
        if (is_rep[r.s] && !is_rep[r.q]) upd(r.qaf >= c_af ? strict : loose, r.qaf >= c_af ? strict_order : loose_order, r.q, r.s, r.ani, r.qaf);
        if (is_rep[r.q] && !is_rep[r.s]) upd(r.saf >= c_af ? strict : loose, r.saf >= c_af ? strict_order : loose_order, r.s, r.q, r.ani, r.saf);
    }
    TmpOut out(out_txt);
    std::string buf = "genome\tnearest_representative_genome\taverage_nucleotide_identity\talignment_fraction\tmatch_category\n";
    for (uint32_t i = 0; i < n_reps; i++) {
        const char *s = nm.shown(reps[i]);
        buf.append(s); buf.push_back('\t'); buf.append(s); buf.append("\t100.0\t100.0\trepresentative_to_self\n");
    }
    auto put = [&](uint32_t g, const Best &b, const char *cat) {
        buf.append(nm.shown(g)); buf.push_back('\t');
        if (b.who.empty()) buf.append("NA");      // an entry that was looked up and never assigned (cannot happen with ANI > 0; kept for the form)
        for (size_t k = 0; k < b.who.size(); k++) { if (k) buf.append(", "); buf.append(nm.shown(b.who[k])); }
        buf.push_back('\t'); buf.append(py_float(b.ani)); buf.push_back('\t'); buf.append(py_float(b.af)); buf.push_back('\t'); buf.append(cat); buf.push_back('\n');
        if (buf.size() > (1u << 22)) { out.put(buf); buf.clear(); }
    };
    for (uint32_t g : strict_order) put(g, strict[g], strict[g].ani >= c_ani ? "within_cutoffs_requested" : "outside_cutoffs_requested");
    for (uint32_t g : loose_order) if (!strict[g].touched) put(g, loose[g], "outside_cutoffs_requested");
    out.put(buf);
    out.commit();
}

int fail(char *err, size_t errlen, const char *what)
{
    if (err && errlen) { strncpy(err, what, errlen - 1); err[errlen - 1] = 0; }
    return 1;
}

}   // namespace

#define SEL_TRY try {
#define SEL_CATCH } catch (const std::bad_alloc &) { return fail(err, errlen, "out of memory"); } \
                    catch (const std::exception &e) { return fail(err, errlen, e.what()); }

extern "C" int skder_amd_select_greedy(const skder_edge_t *rows, uint64_t n_rows, uint32_t n_genomes, const char *const *paths, const uint64_t *n50,
                                       const char *const *display_names, double min_ani_pct, double min_af_pct, const char *info_txt,
                                       const char *sorted_txt, const char *results_txt, uint32_t *reps, uint32_t *n_reps, char *err, size_t errlen)
{
    SEL_TRY
    check_args(rows, n_rows, n_genomes, paths);
    if (!n50) throw SelError("null N50 table");
    greedy(rows, n_rows, n_genomes, Names{paths, display_names}, n50, min_ani_pct, min_af_pct, info_txt, sorted_txt, results_txt, reps, n_reps);
    return 0;
    SEL_CATCH
}

extern "C" int skder_amd_select_dynamic(const skder_edge_t *rows, uint64_t n_rows, uint32_t n_genomes, const char *const *paths, const uint64_t *n50,
                                        const char *const *display_names, double min_ani_pct, double min_af_pct, double max_af_diff_pct,
                                        const char *results_txt, uint32_t *reps, uint32_t *n_reps, char *err, size_t errlen)
{
    SEL_TRY
    check_args(rows, n_rows, n_genomes, paths);
    if (!n50) throw SelError("null N50 table");
    dynamic(rows, n_rows, n_genomes, Names{paths, display_names}, n50, min_ani_pct, min_af_pct, max_af_diff_pct, results_txt, reps, n_reps);
    return 0;
    SEL_CATCH
}

extern "C" int skder_amd_select_clusters(const skder_edge_t *rows, uint64_t n_rows, uint32_t n_genomes, const char *const *paths,
                                         const char *const *display_names, const uint32_t *reps, uint32_t n_reps, double af_cutoff_pct,
                                         double ani_cutoff_pct, const char *clustering_txt, char *err, size_t errlen)
{
    SEL_TRY
    check_args(rows, n_rows, n_genomes, paths);
    if (!reps && n_reps) throw SelError("null representative list");
    if (!clustering_txt) throw SelError("null output name");
    clusters(rows, n_rows, n_genomes, Names{paths, display_names}, reps, n_reps, af_cutoff_pct, ani_cutoff_pct, clustering_txt);
    return 0;
    SEL_CATCH
}

// rows whose TABLE TEXT passes two cut-offs: ani >= ani_cut and column `af_column` (4: Align_fraction_ref, 5: Align_fraction_query) >= af_cut,
// compared as the reference compares the parsed two-decimal text (skder.py:127-129 on a `skani search` table)
extern "C" int skder_amd_rows_pass(const skder_edge_t *rows, uint64_t n_rows, double ani_cut_pct, double af_cut_pct, int af_column, uint8_t *pass)
{
    if ((!rows && n_rows) || !pass || (af_column != 4 && af_column != 5)) return 1;
    const int32_t c_ani = cents_at_least(ani_cut_pct), c_af = cents_at_least(af_cut_pct);
    parallel_ranges(n_rows, [&](unsigned, size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; i++)
            pass[i] = pct2_cents((float)rows[i].ani) >= c_ani && pct2_cents((float)(af_column == 4 ? rows[i].af_ref : rows[i].af_query)) >= c_af;
    });
    return 0;
}

// the rounding rule by itself, for the tests (exhaustive comparison with printf)
extern "C" int64_t skder_amd_pct2_cents(float fraction) { return pct2_cents(fraction); }
