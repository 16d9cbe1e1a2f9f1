// ginflate.hip -- DEFLATE (RFC 1951) on the device: one WAVEFRONT per stream, 1,024 streams in flight on an MI355X.
//
// The reference's real inputs are .fasta.gz (GTDB downloads, its own 34 test genomes); with the FASTA parser on the device the
// host's inflate is what a gzip ingest waits for (16 granted CPUs: ~10 GB/s of text).  A DEFLATE stream is serial, but a
// dereplication job has thousands of them, and the compressed bytes are a third of what crosses PCIe otherwise.
//
// One wavefront decodes one stream, every lane holding the SAME decoder state (bit buffer, positions): table look-ups are
// broadcast LDS reads, the instruction stream is uniform, and the 64 lanes are used where the work is wide --
//   * input: 256 bytes of the stream sit IN REGISTERS (lane k holds word k; the next 256 are requested a block ahead), the bit
//     buffer refills by a cross-lane read -- no memory access on the decoder's critical path;
//   * Huffman tables in LDS (9-bit root for literals / lengths, 6-bit for distances, second-level tables behind them: the
//     sizes zlib proves sufficient, 852 + 592 entries), built per block with the entry fills spread over the lanes;
//   * the 32 KB window in LDS: a match is ONE cooperative copy (lane k moves byte k; overlapping matches read modulo the
//     distance), a literal one byte store;
//   * output: every completed 4 KB of the window goes to HBM in 16-byte stores.
// 40.5 KB of LDS per wavefront: four per CU.  The host parses the gzip header and trailer (it read the file); the CRC-32 of
// the text is computed by a second kernel (crc32_kernel: one wavefront per stream, a slice per lane, slices combined with the
// x^n mod P arithmetic of zlib's crc32_combine).  Every table index, distance, input and output position is checked:
// a damaged stream ends with a status, never with an access outside its buffers.
#include "device_utils.h"
#include "engine.h"

#define GI_WIN 32768u
#define GI_LIT_ROOT 9u
#define GI_DIST_ROOT 6u
#define GI_LIT_N 852u          // zlib: ENOUGH_LENS for a 9-bit root and 286 symbols
#define GI_DIST_N 592u         // zlib: ENOUGH_DISTS for a 6-bit root and 30 symbols

enum : uint32_t { GK_LITERAL = 0, GK_MATCH = 1, GK_END = 2, GK_SUB = 3, GK_INVALID = 4 };
// table entry: [31:16] value, [15:12] kind, [11:8] extra bits (or index bits of a second-level table), [7:0] bits consumed
#define GI_ENTRY(V, K, X, L) ((uint32_t)(V) << 16 | (uint32_t)(K) << 12 | (uint32_t)(X) << 8 | (uint32_t)(L))

__constant__ uint16_t GI_LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__constant__ uint8_t GI_LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__constant__ uint16_t GI_OFF_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__constant__ uint8_t GI_OFF_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__constant__ uint8_t GI_PRE_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

struct __attribute__((aligned(16))) GiLds {
    uint8_t win[GI_WIN];
    uint32_t lit[GI_LIT_N];
    uint32_t dist[GI_DIST_N];
    union {
        struct { uint32_t pre[128]; } in;                                       // the code-length code's table, while a block's code lengths are read
        struct { uint8_t sub_bits[512]; uint16_t sub_start[512]; } build;      // while the literal / distance tables are built
    } u;
    uint8_t lens[352];        // fixed code: 288 + 32; dynamic: the 19 lengths of the code-length code in front, the others from 32 on
    uint32_t count[16];
    uint32_t nc[16];
};
static_assert(sizeof(GiLds) <= 40960, "four wavefronts per CU");

#define GI_UNI(X) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(X)))      // a value that is the same in every lane, as a scalar
#define GI_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

struct GiIn {
    const uint8_t *src;      // the stream
    uint32_t len;            // its bytes
    uint64_t bb;             // bit buffer, next bit of the stream in bit 0
    uint32_t nb;             // valid bits
    uint32_t ipos;           // next input byte that enters the bit buffer (a multiple of 4)
    uint32_t cur, nxt;       // input IN REGISTERS: lane k holds word k of the 256-byte block that contains ipos (cur) and of the one behind it (nxt,
                             // requested a block ahead: its latency is hidden behind ~700 bytes of decoded text)
};

// word `lane` of the 256-byte block b of the stream (bytes behind its end read as zero)
__device__ __forceinline__ uint32_t gi_load_word(const GiIn &I, uint32_t block, uint32_t lane)
{
    const uint64_t at = (uint64_t)block * 256u + lane * 4u;
    uint32_t v = 0;
    if (at + 4u <= I.len) __builtin_memcpy(&v, I.src + at, 4);
    else for (uint32_t k = 0; k < 4u; k++) if (at + k < I.len) v |= (uint32_t)I.src[at + k] << (8u * k);
    return v;
}
__device__ __forceinline__ void gi_start(GiIn &I, uint32_t lane)
{
    I.bb = 0; I.nb = 0; I.ipos = 0;
    I.cur = gi_load_word(I, 0u, lane);
    I.nxt = gi_load_word(I, 1u, lane);
}
// at least 33 valid bits: 32 more from the register block (a cross-lane read: no memory access on the decoder's critical path)
__device__ __forceinline__ void gi_refill(GiIn &I, uint32_t lane)
{
    while (I.nb <= 32u) {
        const uint32_t wsel = (uint32_t)__builtin_amdgcn_readfirstlane((int)((I.ipos >> 2) & 63u));
        const uint32_t word = (uint32_t)__builtin_amdgcn_readlane((int)I.cur, (int)wsel);
        I.bb |= (uint64_t)word << I.nb;
        I.nb += 32u; I.ipos += 4u;
        if ((I.ipos & 255u) == 0u) {                 // into the next block: it is here already; request the one behind it
            I.cur = I.nxt;
            I.nxt = gi_load_word(I, (I.ipos >> 8) + 1u, lane);
        }
    }
}
__device__ __forceinline__ uint32_t gi_take(GiIn &I, uint32_t n)
{
    const uint32_t v = (uint32_t)(I.bb & ((1ull << n) - 1ull));
    I.bb >>= n; I.nb -= n;
    return v;
}

__device__ __forceinline__ uint32_t gi_symbol_entry(int tk, uint32_t sym, uint32_t len)
{
    if (tk == 0) return GI_ENTRY(sym, GK_LITERAL, 0, len);                                   // code-length code
    if (tk == 1) {
        if (sym < 256u) return GI_ENTRY(sym, GK_LITERAL, 0, len);
        if (sym == 256u) return GI_ENTRY(0, GK_END, 0, len);
        if (sym > 285u) return GI_ENTRY(0, GK_INVALID, 0, len);
        return GI_ENTRY(GI_LEN_BASE[sym - 257u], GK_MATCH, GI_LEN_EXTRA[sym - 257u], len);
    }
    if (sym > 29u) return GI_ENTRY(0, GK_INVALID, 0, len);
    return GI_ENTRY(GI_OFF_BASE[sym], GK_MATCH, GI_OFF_EXTRA[sym], len);
}

// canonical Huffman code of lens[0 .. n) (LDS) -> look-up table indexed by the next `root` bits (bit-reversed codes), second-level
// tables behind it for longer codes (gunzip.cpp build_table, which tests/test_gunzip.py holds against zlib).  Uniform over the
// wavefront; the entry fills are spread over the lanes.  false: over-subscribed, incomplete where zlib does not allow it, or
// more entries than the table has.
__device__ bool gi_build(GiLds &L, int tk, const uint8_t *lens, uint32_t n, uint32_t root, uint32_t *table, uint32_t table_size, uint32_t lane, bool &clobbered)
{
    if (lane < 16u) L.count[lane] = 0u;
    GI_SYNC();
    for (uint32_t i = lane; i < n; i += 64u) atomicAdd(&L.count[lens[i]], 1u);
    GI_SYNC();
    uint32_t max_len = 15u;
    while (max_len > 0u && GI_UNI(L.count[max_len]) == 0u) max_len--;
    const uint32_t root_size = 1u << root;
    for (uint32_t i = lane; i < root_size; i += 64u) table[i] = GI_ENTRY(0, GK_INVALID, 0, 1);
    GI_SYNC();
    if (max_len == 0u) return tk != 0;
    int left = 1;
    for (uint32_t len = 1; len <= 15u; len++) { left = left * 2 - (int)GI_UNI(L.count[len]); if (left < 0) return false; }
    if (left > 0 && (tk == 0 || max_len != 1u)) return false;
    {
        uint32_t code = 0, prev = 0;
        for (uint32_t len = 1; len <= 15u; len++) {
            code = (code + prev) << 1;
            prev = GI_UNI(L.count[len]);
            if (lane == 0u) L.nc[len] = code;
        }
        GI_SYNC();
    }
    if (max_len > root) {
        // codes longer than the root: how many bits the second-level table of their root prefix must index.  A first walk over the
        // symbols with a copy of the code counters (in count[], which is not needed any more); this path uses the memory of the input ring
        clobbered = true;
        for (uint32_t i = lane; i < root_size; i += 64u) { L.u.build.sub_bits[i] = 0; L.u.build.sub_start[i] = 0; }
        if (lane < 16u) L.count[lane] = L.nc[lane];
        GI_SYNC();
        for (uint32_t s = 0; s < n; s++) {
            const uint32_t len = GI_UNI(lens[s]);
            if (!len) continue;
            const uint32_t code = GI_UNI(L.count[len]);
            GI_SYNC();
            if (lane == 0u) {
                L.count[len] = code + 1u;
                if (len > root) {
                    const uint32_t prefix = (__brev(code) >> (32u - len)) & (root_size - 1u);
                    if (len - root > L.u.build.sub_bits[prefix]) L.u.build.sub_bits[prefix] = (uint8_t)(len - root);
                }
            }
            GI_SYNC();
        }
        uint32_t next = root_size;
        for (uint32_t pfx = 0; pfx < root_size; pfx++) {
            const uint32_t sb = GI_UNI(L.u.build.sub_bits[pfx]);
            if (!sb) continue;
            const uint32_t size = 1u << sb;
            if (next + size > table_size) return false;
            for (uint32_t i = lane; i < size; i += 64u) table[next + i] = GI_ENTRY(0, GK_INVALID, 0, 1);
            if (lane == 0u) { L.u.build.sub_start[pfx] = (uint16_t)next; table[pfx] = GI_ENTRY(next, GK_SUB, sb, root); }
            next += size;
        }
        GI_SYNC();
    }
    for (uint32_t s = 0; s < n; s++) {
        const uint32_t len = GI_UNI(lens[s]);
        if (!len) continue;
        const uint32_t code = GI_UNI(L.nc[len]);
        GI_SYNC();
        if (lane == 0u) L.nc[len] = code + 1u;
        const uint32_t rev = __brev(code) >> (32u - len);
        if (len <= root) {
            const uint32_t e = gi_symbol_entry(tk, s, len);
            for (uint32_t k = lane; k < (1u << (root - len)); k += 64u) table[rev + (k << len)] = e;
        } else {
            const uint32_t pfx = rev & (root_size - 1u), sb = GI_UNI(L.u.build.sub_bits[pfx]), st = GI_UNI(L.u.build.sub_start[pfx]);
            const uint32_t e = gi_symbol_entry(tk, s, len - root);
            for (uint32_t k = lane; k < (1u << (sb - (len - root))); k += 64u) table[st + (rev >> root) + (k << (len - root))] = e;
        }
        GI_SYNC();
    }
    return true;
}

// statuses (skder_gz_result_t::status)
#define GI_OK 0u
#define GI_CORRUPT 2u
#define GI_TRUNCATED 3u
#define GI_OUTPUT_FULL 4u

struct GiJob { uint64_t in_off; uint32_t in_len, pad; uint64_t out_off, out_cap; };
struct GiResult { uint32_t status, crc, in_used, pad; uint64_t out_len; };

__global__ __launch_bounds__(64) void ginflate_kernel(const uint8_t *__restrict__ in, const GiJob *__restrict__ jobs, uint8_t *__restrict__ out,
                                                      GiResult *__restrict__ res)
{
    __shared__ GiLds L;
    const uint32_t lane = threadIdx.x;
    const GiJob J = jobs[blockIdx.x];
    uint8_t *o = out + J.out_off;
    GiIn I;
    I.src = in + J.in_off; I.len = J.in_len;
    gi_start(I, lane);
    uint64_t w = 0, flushed = 0;          // bytes of text produced; bytes of it in HBM (a multiple of 4096 until the end)
    uint32_t status = GI_OK;
    for (;;) {
        gi_refill(I, lane);
        const uint32_t last = gi_take(I, 1), type = gi_take(I, 2);
        if (type == 0u) {
            gi_take(I, I.nb & 7u);
            gi_refill(I, lane);
            const uint32_t len = gi_take(I, 16), nlen = gi_take(I, 16);
            if ((len ^ 0xFFFFu) != nlen) { status = GI_CORRUPT; break; }
            if (w + len > J.out_cap) { status = GI_OUTPUT_FULL; break; }
            for (uint32_t i = 0; i < len; i++) {
                gi_refill(I, lane);
                const uint32_t b = gi_take(I, 8);
                if (lane == 0u) L.win[(uint32_t)w & (GI_WIN - 1u)] = (uint8_t)b;
                w++;
                if (w - flushed >= 4096u) {
                    GI_SYNC();
                    const uint32_t base = (uint32_t)flushed & (GI_WIN - 1u);
                    for (uint32_t k = lane; k < 256u; k += 64u) *reinterpret_cast<uint4 *>(o + flushed + k * 16u) = *reinterpret_cast<const uint4 *>(L.win + base + k * 16u);
                    flushed += 4096u;
                }
            }
            GI_SYNC();
        } else if (type == 1u || type == 2u) {
            if (type == 1u) {
                for (uint32_t i = lane; i < 288u; i += 64u) L.lens[i] = i < 144u ? 8 : i < 256u ? 9 : i < 280u ? 7 : 8;
                if (lane < 32u) L.lens[288u + lane] = 5;
                GI_SYNC();
                bool clob = false;
                if (!gi_build(L, 1, L.lens, 288u, GI_LIT_ROOT, L.lit, GI_LIT_N, lane, clob) || !gi_build(L, 2, L.lens + 288, 32u, GI_DIST_ROOT, L.dist, GI_DIST_N, lane, clob)) { status = GI_CORRUPT; break; }
            } else {
                const uint32_t nlit = gi_take(I, 5) + 257u, ndist = gi_take(I, 5) + 1u, npre = gi_take(I, 4) + 4u;
                if (nlit > 286u || ndist > 30u) { status = GI_CORRUPT; break; }
                if (lane < 19u) L.lens[lane] = 0;
                GI_SYNC();
                for (uint32_t i = 0; i < npre; i++) {
                    gi_refill(I, lane);
                    const uint32_t v = gi_take(I, 3);
                    if (lane == 0u) L.lens[GI_PRE_ORDER[i]] = (uint8_t)v;
                }
                GI_SYNC();
                // the code-length code's table: built in the scratch behind the ring, so the ring survives (root 7: 128 entries)
                {
                    // gi_build's second-level path would need the ring's memory: the code-length code has at most 7-bit codes, so it never takes it
                    bool never = false;
                    if (!gi_build(L, 0, L.lens, 19u, 7u, L.u.in.pre, 128u, lane, never)) { status = GI_CORRUPT; break; }
                }
                uint32_t i = 0;
                bool bad = false;
                while (i < nlit + ndist) {
                    gi_refill(I, lane);
                    const uint32_t e = GI_UNI(L.u.in.pre[(uint32_t)I.bb & 127u]);
                    if (((e >> 12) & 15u) != GK_LITERAL) { bad = true; break; }
                    gi_take(I, e & 255u);
                    const uint32_t sym = e >> 16;
                    if (sym < 16u) { if (lane == 0u) L.lens[32u + i] = (uint8_t)sym; i++; GI_SYNC(); continue; }
                    uint32_t rep, val = 0;
                    if (sym == 16u) { if (i == 0u) { bad = true; break; } rep = 3u + gi_take(I, 2); val = GI_UNI(L.lens[32u + i - 1u]); }
                    else if (sym == 17u) rep = 3u + gi_take(I, 3);
                    else rep = 11u + gi_take(I, 7);
                    if (i + rep > nlit + ndist) { bad = true; break; }
                    for (uint32_t k = lane; k < rep; k += 64u) L.lens[32u + i + k] = (uint8_t)val;
                    i += rep;
                    GI_SYNC();
                }
                if (bad || L.lens[32u + 256u] == 0) { status = GI_CORRUPT; break; }
                // (the code lengths sit at lens[32 ..]: the 19 lengths of the code-length code kept the front)
                bool clob = false;
                if (!gi_build(L, 1, L.lens + 32, nlit, GI_LIT_ROOT, L.lit, GI_LIT_N, lane, clob) || !gi_build(L, 2, L.lens + 32 + nlit, ndist, GI_DIST_ROOT, L.dist, GI_DIST_N, lane, clob)) { status = GI_CORRUPT; break; }
            }
            // ---- literals, lengths and distances of the block
            uint32_t e_ahead = 0;
            bool ahead = false;
            for (;;) {
                if (I.nb < 20u) gi_refill(I, lane);                       // a literal / length code and its extra bits: 15 + 5
                // (every table entry goes through readfirstlane: the decoder's state is the same in all lanes, and saying so puts the whole
                // serial part -- bit buffer, fields, positions -- on the SCALAR unit, which runs dependent integer code faster than the
                // vector unit's four-cycle cadence)
                uint32_t e = GI_UNI(ahead ? e_ahead : L.lit[(uint32_t)I.bb & ((1u << GI_LIT_ROOT) - 1u)]);
                ahead = false;
                if (((e >> 12) & 15u) == GK_SUB) { gi_take(I, GI_LIT_ROOT); const uint32_t ix = (e >> 16) + ((uint32_t)I.bb & ((1u << ((e >> 8) & 15u)) - 1u)); e = GI_UNI(L.lit[ix < GI_LIT_N ? ix : 0u]); }
                gi_take(I, e & 255u);
                const uint32_t kind = (e >> 12) & 15u;
                if (kind == GK_LITERAL) {
                    if (w >= J.out_cap) { status = GI_OUTPUT_FULL; break; }
                    if (lane == 0u) L.win[(uint32_t)w & (GI_WIN - 1u)] = (uint8_t)(e >> 16);
                    w++;
                } else if (kind == GK_MATCH) {
                    const uint32_t len = (e >> 16) + gi_take(I, (e >> 8) & 15u);
                    if (I.nb < 28u) gi_refill(I, lane);                   // a distance code and its extra bits: 15 + 13
                    uint32_t d = GI_UNI(L.dist[(uint32_t)I.bb & ((1u << GI_DIST_ROOT) - 1u)]);
                    if (((d >> 12) & 15u) == GK_SUB) { gi_take(I, GI_DIST_ROOT); const uint32_t ix = (d >> 16) + ((uint32_t)I.bb & ((1u << ((d >> 8) & 15u)) - 1u)); d = GI_UNI(L.dist[ix < GI_DIST_N ? ix : 0u]); }
                    if (((d >> 12) & 15u) != GK_MATCH) { status = GI_CORRUPT; break; }
                    gi_take(I, d & 255u);
                    const uint32_t dist = (d >> 16) + gi_take(I, (d >> 8) & 15u);
                    if ((uint64_t)dist > w) { status = GI_CORRUPT; break; }
                    if (w + len > J.out_cap) { status = GI_OUTPUT_FULL; break; }
                    GI_SYNC();
                    // the next symbol's table entry is requested BEFORE the copy: it depends on the bit buffer only, and the copy's
                    // read-then-write round trip through LDS is the longest wait of a match
                    e_ahead = L.lit[(uint32_t)I.bb & ((1u << GI_LIT_ROOT) - 1u)];
                    ahead = I.nb >= 20u;
                    const uint32_t from = (uint32_t)(w - dist), to = (uint32_t)w;
                    if (dist >= len) {
                        for (uint32_t k = lane; k < len; k += 64u) L.win[(to + k) & (GI_WIN - 1u)] = L.win[(from + k) & (GI_WIN - 1u)];
                    } else {
                        uint32_t b[5];            // overlapping: the bytes of [w - dist, w) repeat; read all before writing any (len <= 258: five trips)
#pragma unroll
                        for (uint32_t t = 0; t < 5u; t++) { const uint32_t k = lane + 64u * t; b[t] = k < len ? L.win[(from + k % dist) & (GI_WIN - 1u)] : 0u; }
                        GI_SYNC();
#pragma unroll
                        for (uint32_t t = 0; t < 5u; t++) { const uint32_t k = lane + 64u * t; if (k < len) L.win[(to + k) & (GI_WIN - 1u)] = (uint8_t)b[t]; }
                    }
                    w += len;
                } else if (kind == GK_END) {
                    break;
                } else { status = GI_CORRUPT; break; }
                if (w - flushed >= 4096u) {
                    GI_SYNC();
                    const uint32_t base = (uint32_t)flushed & (GI_WIN - 1u);
                    for (uint32_t k = lane; k < 256u; k += 64u) *reinterpret_cast<uint4 *>(o + flushed + k * 16u) = *reinterpret_cast<const uint4 *>(L.win + base + k * 16u);
                    flushed += 4096u;
                }
            }
            if (status != GI_OK) break;
        } else { status = GI_CORRUPT; break; }
        if (last) break;
    }
    GI_SYNC();
    if (status == GI_OK) {
        for (uint64_t k = flushed + lane; k < w; k += 64u) o[k] = L.win[(uint32_t)k & (GI_WIN - 1u)];
        // the stream ends inside the input: bits that were never there read as zero and may have decoded to anything
        const uint32_t used = I.ipos - (I.nb >> 3);
        if (used > I.len) status = GI_TRUNCATED;
        if (lane == 0u) { res[blockIdx.x].in_used = used; }
    }
    if (lane == 0u) { res[blockIdx.x].status = status; res[blockIdx.x].out_len = w; }
}

// ---- CRC-32 (IEEE 802.3, reflected) of every stream's text: one wavefront per stream, a contiguous slice per lane (bytewise table
// look-ups), the 64 slice CRCs combined pairwise with zlib's crc32_combine arithmetic: crc(A || B) = x^(8 |B|) * crc(A) + crc(B) mod P
#define GI_POLY 0xEDB88320u
__device__ __forceinline__ uint32_t gi_multmodp(uint32_t a, uint32_t b)
{
    uint32_t m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) { p ^= b; if ((a & (m - 1u)) == 0u) break; }
        m >>= 1;
        b = (b & 1u) ? (b >> 1) ^ GI_POLY : b >> 1;
    }
    return p;
}
// x^(8 n) mod P
__device__ __forceinline__ uint32_t gi_x8nmodp(uint64_t n)
{
    uint32_t p = 1u << 31, sq = 0x00800000u;        // x^0; x^8 (reflected: x^k is the bit 31 - k)
    while (n) {
        if (n & 1u) p = gi_multmodp(sq, p);
        sq = gi_multmodp(sq, sq);
        n >>= 1;
    }
    return p;
}
__global__ __launch_bounds__(64) void crc32_kernel(const uint8_t *__restrict__ out, const GiJob *__restrict__ jobs, GiResult *__restrict__ res)
{
    __shared__ uint32_t T[256];
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < 256u; i += 64u) {
        uint32_t c = i;
        for (int k = 0; k < 8; k++) c = (c >> 1) ^ (GI_POLY & (0u - (c & 1u)));
        T[i] = c;
    }
    __syncthreads();
    const uint64_t n = res[blockIdx.x].out_len;
    const uint8_t *p = out + jobs[blockIdx.x].out_off;
    const uint64_t per = (n + 63u) / 64u;
    const uint64_t lo = per * lane < n ? per * lane : n, hi = lo + per < n ? lo + per : n;
    uint32_t c = 0;                                 // raw CRC state of the slice with a zero initial value (combinable)
    for (uint64_t k = lo; k < hi; k++) c = (c >> 8) ^ T[(c ^ p[k]) & 255u];
    uint64_t len = hi - lo;
    for (uint32_t step = 1; step < 64u; step <<= 1) {
        const uint32_t c2 = (uint32_t)__shfl_down((int)c, (int)step, 64);
        const uint64_t l2 = (uint64_t)__shfl_down((long long)len, (int)step, 64);
        // (every lane computes; lanes whose partner is beyond the wavefront get garbage that nobody reads)
        c = gi_multmodp(gi_x8nmodp(l2), c) ^ c2;
        len += l2;
    }
    // the whole text with the standard pre- and post-conditioning: crc(0xFFFFFFFF-initialised) = combine(crc of n zero-prefixed ...)
    if (lane == 0u) {
        // initial value 0xFFFFFFFF = the CRC state after "virtual" bytes: fold it in as x^(8 n) * 0xFFFFFFFF
        const uint32_t full = gi_multmodp(gi_x8nmodp(n), 0xFFFFFFFFu) ^ c;
        res[blockIdx.x].crc = full ^ 0xFFFFFFFFu;
    }
}

// n streams resident in d_in -> their texts in d_out and the CRC-32 of each, on `st` (asynchronous: the results are in d_results when the
// stream reaches that point).  d_jobs / d_results: device arrays of n entries.
void ginflate_enqueue(hipStream_t st, const uint8_t *d_in, const skder_gz_job_t *d_jobs, uint32_t n, uint8_t *d_out, skder_gz_result_t *d_results)
{
    static_assert(sizeof(GiJob) == sizeof(skder_gz_job_t) && sizeof(GiResult) == sizeof(skder_gz_result_t), "job / result layout");
    if (!n) return;
    hipLaunchKernelGGL(ginflate_kernel, dim3(n), dim3(64), 0, st, d_in, reinterpret_cast<const GiJob *>(d_jobs), d_out, reinterpret_cast<GiResult *>(d_results));
    hipLaunchKernelGGL(crc32_kernel, dim3(n), dim3(64), 0, st, d_out, reinterpret_cast<const GiJob *>(d_jobs), reinterpret_cast<GiResult *>(d_results));
    HIPCHECK(hipGetLastError());
}

// jobs / results: host arrays.  d_in: the DEFLATE streams (raw: behind the gzip header), d_out: the text regions
extern "C" int skder_amd_inflate_device(skder_ctx_t *ctx, const uint8_t *d_in, const skder_gz_job_t *jobs, uint32_t n, uint8_t *d_out,
                                        skder_gz_result_t *results, float *kernel_ms)
{
    if (!ctx || !d_in || !jobs || !d_out || !results) return 1;
    try {
        HIPCHECK(hipSetDevice(ctx->device));
        hipStream_t st = ctx->stream;
        DevBuf<GiJob> dj;
        DevBuf<GiResult> dr;
        dj.resize(n + 1, st); dr.resize(n + 1, st);
        HIPCHECK(hipMemcpyAsync(dj.p, jobs, (size_t)n * sizeof(GiJob), hipMemcpyHostToDevice, st));
        HIPCHECK(hipMemsetAsync(dr.p, 0, (size_t)n * sizeof(GiResult), st));
        HIPCHECK(hipEventRecord(ctx->ev[11], st));
        if (n) hipLaunchKernelGGL(ginflate_kernel, dim3(n), dim3(64), 0, st, d_in, dj.p, d_out, dr.p);
        HIPCHECK(hipEventRecord(ctx->ev[12], st));
        if (n) hipLaunchKernelGGL(crc32_kernel, dim3(n), dim3(64), 0, st, d_out, dj.p, dr.p);
        HIPCHECK(hipEventRecord(ctx->ev[13], st));
        HIPCHECK(hipGetLastError());
        HIPCHECK(hipMemcpyAsync(results, dr.p, (size_t)n * sizeof(GiResult), hipMemcpyDeviceToHost, st));
        HIPCHECK(hipStreamSynchronize(st));
        if (kernel_ms) {
            HIPCHECK(hipEventElapsedTime(&kernel_ms[0], ctx->ev[11], ctx->ev[12]));
            HIPCHECK(hipEventElapsedTime(&kernel_ms[1], ctx->ev[12], ctx->ev[13]));
        }
        return 0;
    } catch (const std::exception &e) { ctx->last_error = e.what(); return 2; }
}
