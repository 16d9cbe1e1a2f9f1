// host_parser_harness.cpp -- CPU-only check of the FASTA parser (skder_amd/csrc/host_io.hip) under
// AddressSanitizer / UBSan: every file is parsed into memory of its own and into a heap block of exactly
// the region size the ingest path would reserve; layouts, record tables and N50 must agree.
// Built and run by tests/test_host_parser_asan.py (file names on the command line).
#include "host_io.h"
#include <cstdio>
#include <sys/stat.h>
int main(int argc, char **argv) {
    for (int fi = 1; fi < argc; fi++) {
        const char *f = argv[fi];
        HostGenome a, b;
        try { read_fasta(f, a); }
        catch (const std::exception &e) { printf("%-12s ERRORED %s\n", f, e.what()); continue; }    // (a truncated gzip file must be refused)
        struct stat sb; stat(f, &sb);
        size_t cap = (((size_t)sb.st_size + sb.st_size / 15 + 256) + 31) & ~(size_t)31;
        std::string name(f);
        bool gz = name.size() > 3 && name.substr(name.size() - 3) == ".gz";
        size_t rc = gz ? a.packed_size + (2u << 20) : cap; uint8_t *region = (uint8_t *)malloc(rc);
        read_fasta(f, b, region, rc);
        bool same = a.packed_size == b.packed_size && a.rec_len == b.rec_len && a.rec_rel == b.rec_rel && a.n50 == b.n50 &&
                    memcmp(a.own, region, a.packed_size) == 0;
        size_t tot = 0; for (auto l : a.rec_len) tot += l;
        printf("%-12s records %zu bases %zu packed %llu n50 %llu first '%s' same %d\n", f, a.rec_len.size(), tot, (unsigned long long)a.packed_size,
               (unsigned long long)a.n50, a.first_name.c_str(), (int)same);
        free(region);
    }
    return 0;
}
