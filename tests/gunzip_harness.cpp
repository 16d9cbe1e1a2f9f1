// tests/gunzip_harness.cpp -- skder_amd/csrc/gunzip.cpp against zlib (test infrastructure; built with ASan + UBSan by tests/test_gunzip.py)
//   gunzip_harness check FILE.gz [CAP]   decode with both; prints "same N" / "both-fail" / "MISMATCH ..."
//   gunzip_harness fuzz FILE.gz SEED N   N damaged copies of the file (bit flips, cuts, splices): the decoder must fail or agree with zlib, never crash
//   gunzip_harness crc SEED N            gunzip_crc32 against zlib's crc32 on random lengths and alignments
//   gunzip_harness time FILE.gz REPS     MB/s of text for both
#include "../skder_amd/csrc/gunzip.h"

#include <zlib.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

static std::vector<uint8_t> slurp(const char *path)
{
    std::vector<uint8_t> v;
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    uint8_t buf[65536];
    size_t k;
    while ((k = fread(buf, 1, sizeof buf, f)) > 0) v.insert(v.end(), buf, buf + k);
    fclose(f);
    return v;
}

// zlib's view of the same bytes: every member, trailing garbage ignored; false = error
static bool zlib_gunzip(const std::vector<uint8_t> &in, std::vector<uint8_t> &out)
{
    out.clear();
    z_stream z;
    memset(&z, 0, sizeof z);
    if (inflateInit2(&z, 15 + 16) != Z_OK) return false;
    z.next_in = const_cast<Bytef *>(in.data()); z.avail_in = (uInt)in.size();
    std::vector<uint8_t> buf(1 << 16);
    bool ok = true, any = false;
    for (;;) {
        z.next_out = buf.data(); z.avail_out = (uInt)buf.size();
        const int rc = inflate(&z, Z_NO_FLUSH);
        out.insert(out.end(), buf.data(), buf.data() + (buf.size() - z.avail_out));
        if (rc == Z_STREAM_END) {
            any = true;
            if (z.avail_in >= 2 && z.next_in[0] == 0x1f && z.next_in[1] == 0x8b) { inflateReset(&z); continue; }
            break;
        }
        if (rc != Z_OK) { ok = false; break; }
        if (z.avail_in == 0 && z.avail_out != 0) { ok = false; break; }     // truncated
    }
    inflateEnd(&z);
    return ok && any;
}

static int check(const std::vector<uint8_t> &in, size_t cap_limit, bool verbose)
{
    std::vector<uint8_t> want;
    const bool zok = zlib_gunzip(in, want);
    // exact-size input and output buffers: the sanitizer sees any access outside them
    uint8_t *inb = (uint8_t *)malloc(in.size() ? in.size() : 1);
    if (!in.empty()) memcpy(inb, in.data(), in.size());
    size_t cap = zok ? want.size() : in.size() * 8 + 1024;
    if (cap_limit != (size_t)-1) cap = cap_limit;
    uint8_t *outb = (uint8_t *)malloc(cap ? cap : 1);
    size_t n = 0;
    const GunzipStatus st = gunzip_buffer(inb, in.size(), outb, cap, &n);
    int rc = 0;
    if (zok && cap >= want.size()) {
        if (st != GUNZIP_OK || n != want.size() || (n && memcmp(outb, want.data(), n) != 0)) {
            printf("MISMATCH zlib ok (%zu bytes), decoder %s (%zu bytes)\n", want.size(), gunzip_status_text(st), n);
            rc = 1;
        } else if (verbose) printf("same %zu\n", n);
    } else if (zok) {
        if (st != GUNZIP_OUTPUT_FULL) { printf("MISMATCH buffer of %zu for %zu bytes: %s\n", cap, want.size(), gunzip_status_text(st)); rc = 1; }
        else if (verbose) printf("full\n");
    } else {
        if (st == GUNZIP_OK) { printf("MISMATCH zlib fails, decoder ok (%zu bytes)\n", n); rc = 1; }
        else if (verbose) printf("both-fail %s\n", gunzip_status_text(st));
    }
    free(inb); free(outb);
    return rc;
}

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    const std::string mode = argv[1];
    if (mode == "check") {
        const auto in = slurp(argv[2]);
        return check(in, argc > 3 ? (size_t)atoll(argv[3]) : (size_t)-1, true);
    }
    if (mode == "fuzz") {
        const auto in = slurp(argv[2]);
        std::mt19937_64 rng((uint64_t)atoll(argv[3]));
        const int n = atoi(argv[4]);
        int bad = 0;
        for (int i = 0; i < n; i++) {
            std::vector<uint8_t> v = in;
            const int kind = (int)(rng() % 6);
            if (kind == 0 && !v.empty()) v.resize(rng() % v.size());                                    // cut
            else if (kind == 1) for (int k = 0; k < 1 + (int)(rng() % 3); k++) v[rng() % v.size()] ^= (uint8_t)(1u << (rng() % 8));   // bit flips
            else if (kind == 2) for (int k = 0; k < 8; k++) v[rng() % v.size()] = (uint8_t)rng();      // bytes
            else if (kind == 3) { size_t a = rng() % v.size(), b = rng() % v.size(); if (a > b) std::swap(a, b); v.erase(v.begin() + a, v.begin() + b); }   // splice out
            else if (kind == 4) { size_t a = rng() % v.size(); v.insert(v.begin() + a, (size_t)(rng() % 64), (uint8_t)rng()); }                             // insert
            else { size_t a = 10 + rng() % 64; if (a < v.size()) v[a] ^= (uint8_t)(1u << (rng() % 8)); }                                                     // near the block header
            { int b = check(v, (size_t)-1, false); if (b && bad < 3) printf("  case %d kind %d\n", i, kind); bad += b; }
        }
        printf("fuzz: %d cases, %d mismatches\n", n, bad);
        return bad ? 1 : 0;
    }
    if (mode == "crc") {
        std::mt19937_64 rng((uint64_t)atoll(argv[2]));
        const int n = atoi(argv[3]);
        std::vector<uint8_t> buf(1 << 16);
        for (auto &b : buf) b = (uint8_t)rng();
        int bad = 0;
        for (int i = 0; i < n; i++) {
            const size_t off = rng() % 64, len = rng() % (i % 4 == 0 ? 200 : buf.size() - 64);
            const uint32_t init = i % 3 == 0 ? 0u : (uint32_t)rng();
            const uint32_t a = gunzip_crc32(init, buf.data() + off, len), b = (uint32_t)crc32(init, buf.data() + off, (uInt)len);
            if (a != b) { if (bad < 5) printf("MISMATCH crc len %zu off %zu: %08x vs %08x\n", len, off, a, b); bad++; }
        }
        printf("crc: %d cases, %d mismatches\n", n, bad);
        return bad ? 1 : 0;
    }
    if (mode == "time") {
        const auto in = slurp(argv[2]);
        const int reps = atoi(argv[3]);
        std::vector<uint8_t> want;
        if (!zlib_gunzip(in, want)) return 1;
        std::vector<uint8_t> out(want.size());
        size_t n = 0;
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < reps; i++) if (gunzip_buffer(in.data(), in.size(), out.data(), out.size(), &n) != GUNZIP_OK) return 1;
        auto t1 = std::chrono::steady_clock::now();
        for (int i = 0; i < reps; i++) zlib_gunzip(in, want);
        auto t2 = std::chrono::steady_clock::now();
        const double a = std::chrono::duration<double>(t1 - t0).count(), b = std::chrono::duration<double>(t2 - t1).count();
        printf("text %zu bytes: decoder %.0f MB/s, zlib %.0f MB/s\n", n, n * 1e-6 * reps / a, n * 1e-6 * reps / b);
        return 0;
    }
    return 2;
}
