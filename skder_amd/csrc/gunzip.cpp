// gunzip.cpp -- whole-buffer gzip decoder of the ingest (RFC 1951 / 1952), host code.
//
// Why it exists: the reference's real inputs are .fasta.gz, the device parses FASTA at PCIe speed (fasta.hip), and with zlib's
// streaming inflate the 16 CPUs an MI355X box grants decode 5 GB/s of text between them -- the whole ingest waits for that.
// The ingest has the compressed file in memory and knows where the text goes (a region of the pinned staging buffer sized from
// the gzip trailer), so this decoder is the simple case: one input buffer, one output buffer that is its own window, no
// streaming state.  What makes it faster than a streaming inflate on DNA text (short Huffman codes, mostly literals):
//   * a 64-bit bit buffer refilled with one unaligned 8-byte load, without a branch;
//   * 11-bit first-level tables for literals/lengths (every code of FASTA text fits), 8-bit for distances, two-level beyond;
//   * one refill per match (or per three literals); matches copied eight bytes at a time, the first sixteen without a test;
//   * the CRC-32 of the trailer by carry-less multiplication (PCLMULQDQ), 16 bytes per step.
// Every read is bounds-checked against the input buffer and every write against the output buffer: the fast loop runs only
// while 8 input bytes and 320 output bytes are left, a careful loop finishes (tests/test_gunzip.py holds the decoder against
// zlib on every block type, header option and a few thousand corrupted streams under ASan + UBSan).
#include "gunzip.h"

#include <cstring>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace {

constexpr unsigned LIT_ROOT = 11, OFF_ROOT = 8, PRE_ROOT = 7;
constexpr unsigned LIT_SIZE = (1u << LIT_ROOT) + 288 * 16, OFF_SIZE = (1u << OFF_ROOT) + 32 * 128, PRE_SIZE = 1u << PRE_ROOT;

// table entry: [31:16] value (literal, length base, distance base, first entry of a second-level table)
//              [15:12] kind, [11:8] extra bits (or bits of the second-level index), [7:0] bits this lookup consumes
enum : uint32_t { K_LITERAL = 0, K_MATCH = 1, K_END = 2, K_SUB = 3, K_INVALID = 4 };
inline uint32_t entry(uint32_t value, uint32_t kind, uint32_t extra, uint32_t len) { return value << 16 | kind << 12 | extra << 8 | len; }
inline uint32_t e_value(uint32_t e) { return e >> 16; }
inline uint32_t e_kind(uint32_t e) { return (e >> 12) & 15u; }
inline uint32_t e_extra(uint32_t e) { return (e >> 8) & 15u; }
inline uint32_t e_len(uint32_t e) { return e & 255u; }

const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t OFF_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t OFF_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
const uint8_t PRE_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

enum TableKind { T_PRE, T_LIT, T_OFF };

inline uint32_t symbol_entry(TableKind tk, unsigned sym, unsigned len)
{
    if (tk == T_PRE) return entry(sym, K_LITERAL, 0, len);
    if (tk == T_LIT) {
        if (sym < 256) return entry(sym, K_LITERAL, 0, len);
        if (sym == 256) return entry(0, K_END, 0, len);
        if (sym > 285) return entry(0, K_INVALID, 0, len);              // 286, 287: in the fixed code, never valid
        return entry(LEN_BASE[sym - 257], K_MATCH, LEN_EXTRA[sym - 257], len);
    }
    if (sym > 29) return entry(0, K_INVALID, 0, len);                   // 30, 31: in the fixed code, never valid
    return entry(OFF_BASE[sym], K_MATCH, OFF_EXTRA[sym], len);
}

inline unsigned reverse_bits(unsigned code, unsigned len)
{
    unsigned r = 0;
    for (unsigned i = 0; i < len; i++) { r = r << 1 | (code & 1u); code >>= 1; }
    return r;
}

// canonical Huffman code of lens[0 .. n) -> lookup table indexed by the next `root` bits of the stream (bit-reversed codes),
// second-level tables behind it for longer codes.  false: over-subscribed, or incomplete where RFC 1951 / zlib do not allow it
// (an incomplete code is accepted only if it is ONE code of one bit; its unused entries are invalid and end the decode).
bool build_table(TableKind tk, const uint8_t *lens, unsigned n, unsigned root, uint32_t *table, unsigned table_size)
{
    unsigned count[16] = {0};
    for (unsigned i = 0; i < n; i++) count[lens[i]]++;
    unsigned max_len = 15;
    while (max_len > 0 && count[max_len] == 0) max_len--;
    const unsigned root_size = 1u << root;
    for (unsigned i = 0; i < root_size; i++) table[i] = entry(0, K_INVALID, 0, 1);
    if (max_len == 0) return tk != T_PRE;                               // no codes: any use of the table is an error
    int left = 1;
    for (unsigned len = 1; len <= 15; len++) {
        left = left * 2 - (int)count[len];
        if (left < 0) return false;                                     // over-subscribed
    }
    if (left > 0 && (tk == T_PRE || max_len != 1)) return false;        // incomplete
    unsigned next_code[16], code = 0;
    count[0] = 0;
    for (unsigned len = 1; len <= 15; len++) { code = (code + count[len - 1]) << 1; next_code[len] = code; }
    // codes longer than the root: how many bits the second-level table of their root prefix must index
    uint8_t sub_bits[1u << LIT_ROOT];
    uint16_t sub_start[1u << LIT_ROOT];
    if (max_len > root) {
        memset(sub_bits, 0, root_size);
        memset(sub_start, 0, sizeof sub_start);
        unsigned nc[16];
        memcpy(nc, next_code, sizeof nc);
        for (unsigned s = 0; s < n; s++) {
            const unsigned len = lens[s];
            if (len <= root) { if (len) nc[len]++; continue; }
            const unsigned prefix = reverse_bits(nc[len]++, len) & (root_size - 1);
            if (len - root > sub_bits[prefix]) sub_bits[prefix] = (uint8_t)(len - root);
        }
        unsigned next = root_size;
        for (unsigned pfx = 0; pfx < root_size; pfx++) {
            if (!sub_bits[pfx]) continue;
            const unsigned size = 1u << sub_bits[pfx];
            if (next + size > table_size) return false;                 // (cannot happen: the tables hold the worst case)
            sub_start[pfx] = (uint16_t)next;
            for (unsigned i = 0; i < size; i++) table[next + i] = entry(0, K_INVALID, 0, 1);
            table[pfx] = entry(next, K_SUB, sub_bits[pfx], root);
            next += size;
        }
    }
    for (unsigned s = 0; s < n; s++) {
        const unsigned len = lens[s];
        if (!len) continue;
        const unsigned rev = reverse_bits(next_code[len]++, len);
        if (len <= root) {
            const uint32_t e = symbol_entry(tk, s, len);
            for (unsigned i = rev; i < root_size; i += 1u << len) table[i] = e;
        } else {
            const unsigned pfx = rev & (root_size - 1), sb = sub_bits[pfx];
            const uint32_t e = symbol_entry(tk, s, len - root);
            for (unsigned i = rev >> root; i < (1u << sb); i += 1u << (len - root)) table[sub_start[pfx] + i] = e;
        }
    }
    return true;
}

struct Tables {
    uint32_t lit[LIT_SIZE], off[OFF_SIZE];
};

struct FixedTables {
    Tables t;
    FixedTables()
    {
        uint8_t l[288], d[32];
        for (int i = 0; i < 144; i++) l[i] = 8;
        for (int i = 144; i < 256; i++) l[i] = 9;
        for (int i = 256; i < 280; i++) l[i] = 7;
        for (int i = 280; i < 288; i++) l[i] = 8;
        for (int i = 0; i < 32; i++) d[i] = 5;
        build_table(T_LIT, l, 288, LIT_ROOT, t.lit, LIT_SIZE);
        build_table(T_OFF, d, 32, OFF_ROOT, t.off, OFF_SIZE);
    }
};

inline uint64_t load64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }     // (little-endian hosts: x86-64)

struct Stream {
    const uint8_t *p, *in_end;        // next input byte not yet in the bit buffer; end of the input
    uint64_t buf = 0;                 // bit buffer, next bit of the stream in bit 0
    unsigned cnt = 0;                 // valid bits in it
    uint8_t *out, *out_begin, *out_end;
};

// byte-wise refill with the end of the input checked (bits that do not exist read as zero: `cnt` says how many are real)
inline void refill_careful(Stream &s)
{
    while (s.cnt < 56 && s.p < s.in_end) { s.buf |= (uint64_t)*s.p++ << s.cnt; s.cnt += 8; }
}

// bits that follow the current position, real ones only
inline bool need_bits(Stream &s, unsigned n)
{
    if (s.cnt >= n) return true;
    refill_careful(s);
    return s.cnt >= n;
}
inline uint32_t take_bits(Stream &s, unsigned n)
{
    const uint32_t v = (uint32_t)(s.buf & ((1ull << n) - 1ull));
    s.buf >>= n; s.cnt -= n;
    return v;
}

inline void copy_match(uint8_t *dst, unsigned dist, unsigned len)
{
    // byte by byte: overlapping matches repeat their own output
    const uint8_t *src = dst - dist;
    for (unsigned i = 0; i < len; i++) dst[i] = src[i];
}

// literals, lengths and distances of one block with both tables built.  member_begin: where this member's text starts (a match
// cannot reach in front of it)
GunzipStatus decode_block(Stream &s, const Tables &t, const uint8_t *member_begin)
{
    const uint32_t *lit = t.lit, *off = t.off;
    for (;;) {
        // ---- fast loop: 8 input bytes left for its unchecked refill, room for a longest match + the copy's overrun.  ONE refill
        // serves a whole match: 56 bits at least, against 15 + 5 (length code, extra bits) + 15 + 13 (distance code, extra bits)
        while (s.in_end - s.p >= 8 && s.out_end - s.out >= 320) {
            // at most 7 bytes enter, so the buffer never takes more than it has room for: cnt ends up in 56 .. 63
            s.buf |= load64(s.p) << s.cnt;
            s.p += (63 - s.cnt) >> 3;
            s.cnt |= 56;
            uint32_t e = lit[s.buf & ((1u << LIT_ROOT) - 1u)];
            if (e_kind(e) == K_SUB) { s.buf >>= LIT_ROOT; s.cnt -= LIT_ROOT; e = lit[e_value(e) + (s.buf & ((1u << e_extra(e)) - 1u))]; }
            s.buf >>= e_len(e); s.cnt -= e_len(e);
            if (e_kind(e) == K_LITERAL) {
                // up to two more literals from the same refill (3 x 15 bits < 56)
                *s.out++ = (uint8_t)e_value(e);
                e = lit[s.buf & ((1u << LIT_ROOT) - 1u)];
                if (e_kind(e) == K_LITERAL) {
                    s.buf >>= e_len(e); s.cnt -= e_len(e);
                    *s.out++ = (uint8_t)e_value(e);
                    e = lit[s.buf & ((1u << LIT_ROOT) - 1u)];
                    if (e_kind(e) == K_LITERAL) {
                        s.buf >>= e_len(e); s.cnt -= e_len(e);
                        *s.out++ = (uint8_t)e_value(e);
                    }
                }
                continue;                                           // whatever e is now is looked up again behind the next refill
            }
            if (e_kind(e) != K_MATCH) {
                if (e_kind(e) == K_END) return GUNZIP_OK;
                return GUNZIP_CORRUPT;
            }
            const unsigned len = e_value(e) + (unsigned)(s.buf & ((1u << e_extra(e)) - 1u));
            s.buf >>= e_extra(e); s.cnt -= e_extra(e);
            uint32_t d = off[s.buf & ((1u << OFF_ROOT) - 1u)];
            if (e_kind(d) == K_SUB) { s.buf >>= OFF_ROOT; s.cnt -= OFF_ROOT; d = off[e_value(d) + (s.buf & ((1u << e_extra(d)) - 1u))]; }
            if (e_kind(d) != K_MATCH) return GUNZIP_CORRUPT;
            s.buf >>= e_len(d); s.cnt -= e_len(d);
            const unsigned dist = e_value(d) + (unsigned)(s.buf & ((1u << e_extra(d)) - 1u));
            s.buf >>= e_extra(d); s.cnt -= e_extra(d);
            if (dist > (size_t)(s.out - member_begin)) return GUNZIP_CORRUPT;
            uint8_t *dst = s.out;
            s.out += len;
            if (dist >= 8) {
                // eight bytes at a time (a word only reads bytes that are final: dist >= 8); the first two words without a test
                // -- the matches of DNA text are 8 bytes long on average --, up to 15 bytes behind the match: inside the margin
                const uint8_t *src = dst - dist;
                memcpy(dst, src, 8); memcpy(dst + 8, src + 8, 8);
                if (len > 16) {
                    dst += 16; src += 16;
                    do { memcpy(dst, src, 8); dst += 8; src += 8; } while (dst < s.out);
                }
            } else if (dist == 1) {
                memset(dst, dst[-1], len);
            } else {
                copy_match(dst, dist, len);
            }
        }
        // ---- careful step: one symbol with every bound checked (the ends of the buffers; damaged streams)
        refill_careful(s);
        uint32_t e = lit[s.buf & ((1u << LIT_ROOT) - 1u)];
        unsigned used = 0;
        if (e_kind(e) == K_SUB) { used = LIT_ROOT; e = lit[e_value(e) + ((s.buf >> LIT_ROOT) & ((1u << e_extra(e)) - 1u))]; }
        used += e_len(e);
        if (e_kind(e) == K_INVALID) return s.cnt < 15 && s.p >= s.in_end ? GUNZIP_TRUNCATED : GUNZIP_CORRUPT;
        if (used > s.cnt) return GUNZIP_TRUNCATED;
        s.buf >>= used; s.cnt -= used;
        if (e_kind(e) == K_LITERAL) {
            if (s.out >= s.out_end) return GUNZIP_OUTPUT_FULL;
            *s.out++ = (uint8_t)e_value(e);
            continue;
        }
        if (e_kind(e) == K_END) return GUNZIP_OK;
        if (!need_bits(s, e_extra(e))) return GUNZIP_TRUNCATED;
        const unsigned len = e_value(e) + take_bits(s, e_extra(e));
        refill_careful(s);
        uint32_t d = off[s.buf & ((1u << OFF_ROOT) - 1u)];
        used = 0;
        if (e_kind(d) == K_SUB) { used = OFF_ROOT; d = off[e_value(d) + ((s.buf >> OFF_ROOT) & ((1u << e_extra(d)) - 1u))]; }
        used += e_len(d);
        if (e_kind(d) != K_MATCH) return s.cnt < 15 && s.p >= s.in_end ? GUNZIP_TRUNCATED : GUNZIP_CORRUPT;
        if (used > s.cnt) return GUNZIP_TRUNCATED;
        s.buf >>= used; s.cnt -= used;
        if (!need_bits(s, e_extra(d))) return GUNZIP_TRUNCATED;
        const unsigned dist = e_value(d) + take_bits(s, e_extra(d));
        if (dist > (size_t)(s.out - member_begin)) return GUNZIP_CORRUPT;
        if (len > (size_t)(s.out_end - s.out)) return GUNZIP_OUTPUT_FULL;
        copy_match(s.out, dist, len);
        s.out += len;
    }
}

// the DEFLATE stream of one member, from s.p (byte-aligned) to the end of its last block; s.p is byte-aligned behind it again
GunzipStatus inflate_member(Stream &s, Tables &dyn)
{
    static const FixedTables fixed;
    const uint8_t *member_begin = s.out;
    s.buf = 0; s.cnt = 0;
    for (;;) {
        if (!need_bits(s, 3)) return GUNZIP_TRUNCATED;
        const uint32_t last = take_bits(s, 1), type = take_bits(s, 2);
        if (type == 0) {
            // stored: the rest of the current byte is skipped; the bit buffer holds whole bytes behind it
            take_bits(s, s.cnt & 7u);
            if (!need_bits(s, 32)) return GUNZIP_TRUNCATED;
            const uint32_t len = take_bits(s, 16), nlen = take_bits(s, 16);
            if ((len ^ 0xFFFFu) != nlen) return GUNZIP_CORRUPT;
            uint32_t left = len;
            while (left && s.cnt >= 8) {                                // bytes already in the bit buffer
                if (s.out >= s.out_end) return GUNZIP_OUTPUT_FULL;
                *s.out++ = (uint8_t)take_bits(s, 8);
                left--;
            }
            if (s.cnt == 0) s.buf = 0;                                    // (bits of the byte at s.p may sit above cnt: the copy below moves s.p)
            if (left) {
                if ((size_t)(s.in_end - s.p) < left) return GUNZIP_TRUNCATED;
                if ((size_t)(s.out_end - s.out) < left) return GUNZIP_OUTPUT_FULL;
                memcpy(s.out, s.p, left);
                s.out += left; s.p += left;
            }
        } else if (type == 1) {
            const GunzipStatus st = decode_block(s, fixed.t, member_begin);
            if (st != GUNZIP_OK) return st;
        } else if (type == 2) {
            if (!need_bits(s, 14)) return GUNZIP_TRUNCATED;
            const unsigned nlit = take_bits(s, 5) + 257, ndist = take_bits(s, 5) + 1, npre = take_bits(s, 4) + 4;
            if (nlit > 286 || ndist > 30) return GUNZIP_CORRUPT;
            uint8_t pre_lens[19] = {0};
            for (unsigned i = 0; i < npre; i++) {
                if (!need_bits(s, 3)) return GUNZIP_TRUNCATED;
                pre_lens[PRE_ORDER[i]] = (uint8_t)take_bits(s, 3);
            }
            uint32_t pre[PRE_SIZE];
            if (!build_table(T_PRE, pre_lens, 19, PRE_ROOT, pre, PRE_SIZE)) return GUNZIP_CORRUPT;
            uint8_t lens[286 + 30 + 138];
            unsigned i = 0;
            while (i < nlit + ndist) {
                refill_careful(s);
                const uint32_t e = pre[s.buf & (PRE_SIZE - 1u)];
                if (e_kind(e) != K_LITERAL) return s.cnt < 7 && s.p >= s.in_end ? GUNZIP_TRUNCATED : GUNZIP_CORRUPT;
                if (e_len(e) > s.cnt) return GUNZIP_TRUNCATED;
                take_bits(s, e_len(e));
                const unsigned sym = e_value(e);
                if (sym < 16) { lens[i++] = (uint8_t)sym; continue; }
                unsigned rep, val = 0;
                if (sym == 16) {
                    if (i == 0) return GUNZIP_CORRUPT;
                    if (!need_bits(s, 2)) return GUNZIP_TRUNCATED;
                    rep = 3 + take_bits(s, 2); val = lens[i - 1];
                } else if (sym == 17) {
                    if (!need_bits(s, 3)) return GUNZIP_TRUNCATED;
                    rep = 3 + take_bits(s, 3);
                } else {
                    if (!need_bits(s, 7)) return GUNZIP_TRUNCATED;
                    rep = 11 + take_bits(s, 7);
                }
                if (i + rep > nlit + ndist) return GUNZIP_CORRUPT;
                memset(lens + i, (int)val, rep);
                i += rep;
            }
            if (lens[256] == 0) return GUNZIP_CORRUPT;                  // no end-of-block code
            if (!build_table(T_LIT, lens, nlit, LIT_ROOT, dyn.lit, LIT_SIZE)) return GUNZIP_CORRUPT;
            if (!build_table(T_OFF, lens + nlit, ndist, OFF_ROOT, dyn.off, OFF_SIZE)) return GUNZIP_CORRUPT;
            const GunzipStatus st = decode_block(s, dyn, member_begin);
            if (st != GUNZIP_OK) return st;
        } else {
            return GUNZIP_CORRUPT;
        }
        if (last) break;
    }
    // back to bytes: whole bytes still in the bit buffer were not consumed
    take_bits(s, s.cnt & 7u);
    s.p -= s.cnt >> 3;
    s.buf = 0; s.cnt = 0;
    return GUNZIP_OK;
}

// ---- CRC-32 (IEEE 802.3, reflected)
struct CrcTables {
    uint32_t t[8][256];
    CrcTables()
    {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
            t[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; i++)
            for (int k = 1; k < 8; k++) t[k][i] = (t[k - 1][i] >> 8) ^ t[0][t[k - 1][i] & 255u];
    }
};
const CrcTables &crc_tables() { static const CrcTables c; return c; }

// raw state in, raw state out (no inversion)
uint32_t crc_tables_update(uint32_t c, const uint8_t *p, size_t n)
{
    const CrcTables &T = crc_tables();
    while (n >= 8) {
        uint64_t v;
        memcpy(&v, p, 8);
        v ^= c;
        c = T.t[7][v & 255u] ^ T.t[6][(v >> 8) & 255u] ^ T.t[5][(v >> 16) & 255u] ^ T.t[4][(v >> 24) & 255u] ^
            T.t[3][(v >> 32) & 255u] ^ T.t[2][(v >> 40) & 255u] ^ T.t[1][(v >> 48) & 255u] ^ T.t[0][v >> 56];
        p += 8; n -= 8;
    }
    while (n--) c = (c >> 8) ^ T.t[0][(c ^ *p++) & 255u];
    return c;
}

#if defined(__x86_64__)
// Folding by carry-less multiplication ("Fast CRC Computation for Generic Polynomials Using PCLMULQDQ Instruction", Gopal et
// al., Intel 2009): n >= 64, a multiple of 16.  Constants of the reflected IEEE polynomial: x^(4*128+32), x^(4*128-32),
// x^(128+32), x^(128-32), x^64 mod P; P' and mu for the Barrett reduction.
__attribute__((target("pclmul,sse4.1")))
inline __m128i crc_fold(__m128i a, __m128i b, __m128i k)
{
    const __m128i h = _mm_clmulepi64_si128(a, k, 0x11);
    a = _mm_clmulepi64_si128(a, k, 0x00);
    return _mm_xor_si128(_mm_xor_si128(a, h), b);
}

__attribute__((target("pclmul,sse4.1")))
uint32_t crc_pclmul(uint32_t crc, const uint8_t *p, size_t n)
{
    const __m128i k1k2 = _mm_set_epi64x(0x01c6e41596, 0x0154442bd4);
    const __m128i k3k4 = _mm_set_epi64x(0x00ccaa009e, 0x01751997d0);
    const __m128i k5 = _mm_set_epi64x(0, 0x0163cd6124);
    const __m128i poly = _mm_set_epi64x(0x01f7011641, 0x01db710641);
    __m128i x1 = _mm_loadu_si128((const __m128i *)(p + 0)), x2 = _mm_loadu_si128((const __m128i *)(p + 16));
    __m128i x3 = _mm_loadu_si128((const __m128i *)(p + 32)), x4 = _mm_loadu_si128((const __m128i *)(p + 48));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
    p += 64; n -= 64;
    while (n >= 64) {
        __m128i h1 = _mm_clmulepi64_si128(x1, k1k2, 0x11), h2 = _mm_clmulepi64_si128(x2, k1k2, 0x11);
        __m128i h3 = _mm_clmulepi64_si128(x3, k1k2, 0x11), h4 = _mm_clmulepi64_si128(x4, k1k2, 0x11);
        x1 = _mm_clmulepi64_si128(x1, k1k2, 0x00); x2 = _mm_clmulepi64_si128(x2, k1k2, 0x00);
        x3 = _mm_clmulepi64_si128(x3, k1k2, 0x00); x4 = _mm_clmulepi64_si128(x4, k1k2, 0x00);
        x1 = _mm_xor_si128(_mm_xor_si128(x1, h1), _mm_loadu_si128((const __m128i *)(p + 0)));
        x2 = _mm_xor_si128(_mm_xor_si128(x2, h2), _mm_loadu_si128((const __m128i *)(p + 16)));
        x3 = _mm_xor_si128(_mm_xor_si128(x3, h3), _mm_loadu_si128((const __m128i *)(p + 32)));
        x4 = _mm_xor_si128(_mm_xor_si128(x4, h4), _mm_loadu_si128((const __m128i *)(p + 48)));
        p += 64; n -= 64;
    }
    // four accumulators into one
    x1 = crc_fold(x1, x2, k3k4); x1 = crc_fold(x1, x3, k3k4); x1 = crc_fold(x1, x4, k3k4);
    while (n >= 16) {
        x1 = crc_fold(x1, _mm_loadu_si128((const __m128i *)p), k3k4);
        p += 16; n -= 16;
    }
    // 128 -> 64 bits
    const __m128i mask32 = _mm_setr_epi32(~0, 0, ~0, 0);
    __m128i t = _mm_clmulepi64_si128(x1, k3k4, 0x10);
    x1 = _mm_xor_si128(_mm_srli_si128(x1, 8), t);
    t = _mm_srli_si128(x1, 4);
    x1 = _mm_and_si128(x1, mask32);
    x1 = _mm_clmulepi64_si128(x1, k5, 0x00);
    x1 = _mm_xor_si128(x1, t);
    // Barrett reduction to 32 bits
    t = _mm_and_si128(x1, mask32);
    t = _mm_clmulepi64_si128(t, poly, 0x10);
    t = _mm_and_si128(t, mask32);
    t = _mm_clmulepi64_si128(t, poly, 0x00);
    x1 = _mm_xor_si128(x1, t);
    return (uint32_t)_mm_extract_epi32(x1, 1);
}
#endif

GunzipStatus parse_header(const uint8_t *&p, const uint8_t *end)
{
    if (end - p < 2 || p[0] != 0x1f || p[1] != 0x8b) return GUNZIP_NOT_GZIP;
    const uint8_t *start = p;
    if (end - p < 10) return GUNZIP_TRUNCATED;
    if (p[2] != 8) return GUNZIP_CORRUPT;                               // compression method: deflate
    const unsigned flg = p[3];
    if (flg & 0xE0u) return GUNZIP_CORRUPT;                             // reserved flag bits
    p += 10;
    if (flg & 4u) {                                                     // FEXTRA
        if (end - p < 2) return GUNZIP_TRUNCATED;
        const size_t xlen = (size_t)p[0] | (size_t)p[1] << 8;
        p += 2;
        if ((size_t)(end - p) < xlen) return GUNZIP_TRUNCATED;
        p += xlen;
    }
    for (unsigned bit = 8u; bit <= 16u; bit <<= 1) {                    // FNAME, FCOMMENT: zero-terminated
        if (!(flg & bit)) continue;
        const uint8_t *z = (const uint8_t *)memchr(p, 0, (size_t)(end - p));
        if (!z) return GUNZIP_TRUNCATED;
        p = z + 1;
    }
    if (flg & 2u) {                                                     // FHCRC: the low half of the CRC-32 of the header so far
        if (end - p < 2) return GUNZIP_TRUNCATED;
        if ((gunzip_crc32(0, start, (size_t)(p - start)) & 0xFFFFu) != ((uint32_t)p[0] | (uint32_t)p[1] << 8)) return GUNZIP_CORRUPT;
        p += 2;
    }
    return GUNZIP_OK;
}

}   // namespace

uint32_t gunzip_crc32(uint32_t crc, const uint8_t *p, size_t n)
{
    uint32_t c = ~crc;
#if defined(__x86_64__)
    static const bool have_clmul = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
    if (have_clmul && n >= 64) {
        const size_t m = n & ~(size_t)15;
        c = crc_pclmul(c, p, m);
        p += m; n -= m;
    }
#endif
    return ~crc_tables_update(c, p, n);
}

const char *gunzip_status_text(GunzipStatus s)
{
    switch (s) {
    case GUNZIP_OK: return "ok";
    case GUNZIP_NOT_GZIP: return "not a gzip file";
    case GUNZIP_CORRUPT: return "invalid compressed data";
    case GUNZIP_TRUNCATED: return "unexpected end of file";
    default: return "text longer than its buffer";
    }
}

GunzipStatus gunzip_buffer(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap, size_t *out_len)
{
    *out_len = 0;
    const uint8_t *p = in, *end = in + in_len;
    Stream s;
    s.out = s.out_begin = out; s.out_end = out + out_cap;
    s.in_end = end;
    Tables *dyn = new Tables;
    struct Free { Tables *t; ~Free() { delete t; } } guard{dyn};
    bool first = true;
    for (;;) {
        GunzipStatus st = parse_header(p, end);
        if (st == GUNZIP_NOT_GZIP && !first) break;                     // trailing bytes that are no member: ignored
        if (st != GUNZIP_OK) return st;
        first = false;
        s.p = p;
        uint8_t *member_out = s.out;
        st = inflate_member(s, *dyn);
        if (st != GUNZIP_OK) return st;
        p = s.p;
        if (end - p < 8) return GUNZIP_TRUNCATED;
        const uint32_t want_crc = (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24;
        const uint32_t want_len = (uint32_t)p[4] | (uint32_t)p[5] << 8 | (uint32_t)p[6] << 16 | (uint32_t)p[7] << 24;
        p += 8;
        const size_t n = (size_t)(s.out - member_out);
        if ((uint32_t)n != want_len) return GUNZIP_CORRUPT;
        if (gunzip_crc32(0, member_out, n) != want_crc) return GUNZIP_CORRUPT;
        *out_len = (size_t)(s.out - out);
    }
    *out_len = (size_t)(s.out - out);
    return GUNZIP_OK;
}
