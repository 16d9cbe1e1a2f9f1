// host_writer_harness.cpp -- CPU-only check of the edge-table side of the drop-in (skder_amd/csrc/host_io.hip) under
// AddressSanitizer / UBSan and, with a row count on the command line, its timing at BASELINE config 5's size:
//   * triangle_rows_order_inplace must give exactly the rows, in exactly the order, of the simple triangle_rows_ordered
//     (skani's hash-map iteration order) -- dense and sparse rows, filtered and not;
//   * rect_rows_order_inplace (parallel sort + merges) must equal a plain std::sort with the same ordering;
//   * write_rows_tsv (blocks formatted in parallel, written at their offsets) must produce the bytes of a plain fprintf loop.
// Built and run by tests/test_host_writer.py.
#include "host_io.h"
#include <chrono>
#include <cstdio>
#include <random>

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static std::vector<skder_edge_t> make_edges(uint32_t n_genomes, size_t n_edges, uint32_t seed, uint32_t block = 100, uint32_t ani_levels = 150000)
{
    // species blocks of `block` genomes: dense rows inside a block, a few stray pairs across (thinned rows)
    std::mt19937_64 rng(seed);
    std::vector<skder_edge_t> E;
    E.reserve(n_edges);
    std::vector<uint64_t> seen;
    while (E.size() < n_edges) {
        uint32_t a = (uint32_t)(rng() % n_genomes), b;
        if (rng() % 10) b = a / block * block + (uint32_t)(rng() % block); else b = (uint32_t)(rng() % n_genomes);
        if (b >= n_genomes || a == b) continue;
        if (a > b) std::swap(a, b);
        seen.push_back((uint64_t)a << 32 | b);
        skder_edge_t e;
        memset(&e, 0, sizeof e);
        e.ref = a; e.query = b;
        e.ani = 0.85 + (double)(rng() % ani_levels) * (0.15 / ani_levels);   /* few levels: many rows of a search table tie on ANI, Ref decides */ e.af_ref = (double)(rng() % 100001) / 1e5; e.af_query = (double)(rng() % 100001) / 1e5;
        E.push_back(e);
    }
    // one record per pair
    std::sort(E.begin(), E.end(), [](const skder_edge_t &x, const skder_edge_t &y) { return x.ref != y.ref ? x.ref < y.ref : x.query < y.query; });
    E.erase(std::unique(E.begin(), E.end(), [](const skder_edge_t &x, const skder_edge_t &y) { return x.ref == y.ref && x.query == y.query; }), E.end());
    std::shuffle(E.begin(), E.end(), rng);
    return E;
}

static bool same(const std::vector<skder_edge_t> &a, const std::vector<skder_edge_t> &b)
{
    return a.size() == b.size() && (a.empty() || memcmp(a.data(), b.data(), a.size() * sizeof(skder_edge_t)) == 0);
}

int main(int argc, char **argv)
{
    const char *dir = argc > 1 ? argv[1] : ".";
    const size_t big = argc > 2 ? strtoull(argv[2], nullptr, 10) : 0;
    GenomeNames names;
    const uint32_t G = big ? 50000 : 1200;
    for (uint32_t g = 0; g < G; g++) {
        names.path.push_back("/data/genomes/species" + std::to_string(g / 100) + "/GCA_" + std::to_string(900000000u + g * 7919u) + ".1_genomic.fna.gz");
        names.first_name.push_back("NZ_" + std::to_string(g) + ".1 Organism name strain " + std::to_string(g % 97) + ", whole genome shotgun sequence");
        names.n50.push_back(1000 + g);
    }
    int bad = 0;
    for (uint32_t round = 0; round < (big ? 0u : 9u); round++) {
        // rounds 0 - 5: species blocks of 100 genomes, 1 .. 100,000 edges; 6: ONE species (every row dense: 400 genomes, all pairs asked for);
        // 7: the same with 12 ANI levels (ties); 8: 1,200 genomes in blocks of 3 (thousands of tiny rows)
        const std::vector<skder_edge_t> E = round < 6 ? make_edges(G, round == 0 ? 1 : (round == 1 ? 37 : 20000u * round), 17 + round)
                                          : round == 6 ? make_edges(400, 120000, 31, 400)
                                          : round == 7 ? make_edges(400, 120000, 32, 400, 12)
                                                       : make_edges(G, 30000, 33, 3);
        for (double min_af : {0.0, 50.0, 99.5}) {
            std::vector<skder_edge_t> want = triangle_rows_ordered(E, min_af), got(E);
            triangle_rows_order_inplace(got, min_af);
            if (!same(want, got)) { printf("triangle order differs: round %u min_af %.1f (%zu / %zu rows)\n", round, min_af, want.size(), got.size()); bad++; }
            // both forms behind it, explicitly (the dispatcher picks by size and memory): the serial in-place one, the parallel one on 2 and 5 threads
            std::vector<skder_edge_t> ser(E);
            triangle_rows_order_serial(ser, min_af);
            if (!same(want, ser)) { printf("serial triangle order differs: round %u min_af %.1f\n", round, min_af); bad++; }
            for (unsigned T : {2u, 5u, 3u}) {       // (T == 3: the big-table branch -- raw scratch block, parallel copy back)
                std::vector<skder_edge_t> par(E);
                if (!triangle_rows_order_parallel(par, min_af, T, T == 3u ? 0 : (256ull << 20))) { printf("parallel triangle order declined: round %u T %u\n", round, T); bad++; }
                else if (!same(want, par)) { printf("parallel triangle order differs: round %u min_af %.1f T %u (%zu / %zu rows)\n", round, min_af, T, want.size(), par.size()); bad++; }
            }
            std::vector<skder_edge_t> r(E), w(E);
            rect_rows_order_inplace(r, min_af);
            w.erase(std::remove_if(w.begin(), w.end(), [&](const skder_edge_t &e) { float a = (float)e.af_ref, b = (float)e.af_query; return (double)(a > b ? a : b) * 100.0 < min_af; }), w.end());
            std::sort(w.begin(), w.end(), [](const skder_edge_t &a, const skder_edge_t &b) {
                if (a.query != b.query) return a.query < b.query;
                float x = (float)a.ani, y = (float)b.ani;
                if (x != y) return x > y;
                return a.ref < b.ref;
            });
            if (!same(r, w)) { printf("rectangle order differs: round %u min_af %.1f\n", round, min_af); bad++; }
            {
                std::vector<skder_edge_t> rs(E);
                rect_rows_order_serial(rs, min_af);
                if (!same(rs, w)) { printf("serial rectangle order differs: round %u min_af %.1f\n", round, min_af); bad++; }
                for (unsigned T : {2u, 5u, 3u}) {
                    std::vector<skder_edge_t> rp(E);
                    if (!rect_rows_order_parallel(rp, min_af, T, T == 3u ? 0 : (256ull << 20))) { printf("parallel rectangle order declined: round %u T %u\n", round, T); bad++; }
                    else if (!same(rp, w)) { printf("parallel rectangle order differs: round %u min_af %.1f T %u\n", round, min_af, T); bad++; }
                }
            }
            // text
            const std::string out = std::string(dir) + "/t.tsv", ref = std::string(dir) + "/t_ref.tsv";
            write_rows_tsv(out, want.data(), want.size(), names, names);
            FILE *f = fopen(ref.c_str(), "w");
            fputs("Ref_file\tQuery_file\tANI\tAlign_fraction_ref\tAlign_fraction_query\tRef_name\tQuery_name\n", f);
            for (const auto &e : want) {
                float ani = (float)e.ani, afr = (float)e.af_ref, afq = (float)e.af_query;
                fprintf(f, "%s\t%s\t%.2f\t%.2f\t%.2f\t%s\t%s\n", names.path[e.ref].c_str(), names.path[e.query].c_str(), (double)(ani * 100.0f),
                        (double)(afr * 100.0f), (double)(afq * 100.0f), names.first_name[e.ref].c_str(), names.first_name[e.query].c_str());
            }
            fclose(f);
            std::string cmd = "cmp -s '" + out + "' '" + ref + "'";
            if (system(cmd.c_str()) != 0) { printf("table text differs: round %u min_af %.1f\n", round, min_af); bad++; }
        }
    }
    if (big) {
        double t0 = now_s();
        std::vector<skder_edge_t> E = make_edges(G, big + big / 2, 5, 1000);      // (duplicates are dropped: ask for more)
        const size_t n = E.size();
        double t_serial = 0;
        if (getenv("HARNESS_TIME_SERIAL")) {         // the one-thread in-place form on a copy, for the comparison
            std::vector<skder_edge_t> S(E);
            const double a = now_s();
            triangle_rows_order_serial(S, 50.0);
            t_serial = now_s() - a;
        }
        double t1 = now_s();
        triangle_rows_order_inplace(E, 50.0);
        double t2 = now_s();
        const std::string out = std::string(dir) + "/big.tsv";
        write_rows_tsv(out, E.data(), E.size(), names, names);
        double t3 = now_s();
        FILE *f = fopen(out.c_str(), "rb");
        fseek(f, 0, SEEK_END);
        const long long bytes = ftell(f);
        fclose(f);
        remove(out.c_str());
        printf("{\"edges\": %zu, \"rows_kept\": %zu, \"threads\": %u, \"order_in_place_s\": %.3f, \"write_s\": %.3f, \"table_bytes\": %lld, \"rows_per_s_write\": %.0f, \"generate_s\": %.3f, \"order_serial_s\": %.3f}\n",
               n, E.size(), ingest_threads(), t2 - t1, t3 - t2, bytes, E.size() / (t3 - t2), t1 - t0 - t_serial, t_serial);
    }
    printf("%s\n", bad ? "FAILED" : "ok");
    return bad ? 1 : 0;
}
