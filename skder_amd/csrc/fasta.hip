// fasta.hip -- FASTA text -> device layout, ON the device (gfx950).
//
// The ingest of the drop-in entry points (host_io.hip) used to scan every file on the host: find the lines, drop headers and
// line ends, copy the bases of the records of >= 500 bases into the 32-byte-aligned layout sketch_tiles_kernel reads, count
// record lengths for N50 (util.n50_calc, /root/reference/src/skDER/util.py:686-724).  Here the host only READS (or inflates)
// the file into a pinned buffer; the text goes over PCIe as it is and the device does the rest.  Two parsers, same results:
// the TILED one further down (three kernels, a wavefront per 4 KB of text: the one the ingest uses) and the one it grew out of:
//
//   one WAVEFRONT per file (fasta_parse_kernel; SKDER_AMD_FASTA_WAVE=1), 4 KB of text per round (64 bytes per lane), state carried from round to round in wave-uniform
//   registers.  Per round: every lane classifies its 64 bytes (line start / header / base / line end), the lanes agree through
//   ballots and three wave scans on (a) whether a lane's chunk starts inside a header line, (b) how many bases of the record
//   that is open at its chunk start precede the chunk, (c) where in the output the records begin -- a record is kept, padded
//   with 'A' to the next 32-byte boundary and entered in the record table iff it has >= 500 bases, which is known when the next
//   header (or the end of the file) closes it; the bases of a record that closes inside the round are written only if it is
//   kept, those of the record still open at the end of the round optimistically (a record that later turns out short is simply
//   overwritten by the next one: the rounds are sequential).  Record lengths of ALL records, and of text in front of the first
//   header, go to a per-file list the host computes the N50 from.
//
// Semantics are read_fasta's (host_io.hip), which the parity tests hold both paths to.  What the kernel does NOT do, it says:
// a file with blanks inside sequence lines (' ', tabs; a '\r' that is not followed by '\n'), with more records than its table
// holds, or larger than 4 GB sets a flag and is parsed by the host instead.
#include "common.h"
#include "device_utils.h"
#include "fasta.h"

#define FA_CHUNK 64u                    // bytes per lane and round
#define FA_ROUND (64u * FA_CHUNK)       // bytes per round

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v)
{
    const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = (uint32_t)__shfl_up((int)v, o, 64);
        if (lane >= (uint32_t)o) v += y;
    }
    return v;
}

// 0x80 in every byte of x that is 0
__device__ __forceinline__ uint32_t zero_bytes(uint32_t x) { return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu); }

// 64 bytes in 16 words -> one bit per byte: f(word) has bit 7 of every byte it selects set; the four bits are gathered to the
// top nibble (two shift-ors) and shifted into place word by word
template <class Fn>
__device__ __forceinline__ uint64_t chunk_mask(const uint32_t (&wd)[16], Fn f)
{
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        uint32_t a = f(wd[k]), b = f(wd[8 + k]);
        a |= a << 7; a |= a << 14;
        b |= b << 7; b |= b << 14;
        lo = (lo >> 4) | (a & 0xF0000000u);
        hi = (hi >> 4) | (b & 0xF0000000u);
    }
    return (uint64_t)hi << 32 | lo;
}

// the lane's 64 bytes at text offset c0 (bytes behind the end of the file read as line ends: the buffer holds '\n' there, a
// chunk that begins behind the end is not loaded at all)
__device__ __forceinline__ void load_chunk(uint32_t (&wd)[16], const uint8_t *tx, uint64_t c0, uint64_t text_len)
{
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint4 v = make_uint4(0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au);
        if (c0 < text_len) v = *reinterpret_cast<const uint4 *>(tx + c0 + 16u * k);
        wd[4 * k] = v.x; wd[4 * k + 1] = v.y; wd[4 * k + 2] = v.z; wd[4 * k + 3] = v.w;
    }
}

// highest set bit of m strictly below `lane`, or -1
__device__ __forceinline__ int prev_set_below(unsigned long long m, uint32_t lane)
{
    const unsigned long long below = m & ((1ull << lane) - 1ull);
    return below ? 63 - __clzll((long long)below) : -1;
}
// lowest set bit of m at or above `lane`, or -1
__device__ __forceinline__ int next_set_from(unsigned long long m, uint32_t lane)
{
    const unsigned long long from = m & ~((1ull << lane) - 1ull);
    return from ? __ffsll((long long)from) - 1 : -1;
}

__global__ __launch_bounds__(64) void fasta_parse_kernel(const uint8_t *__restrict__ text, const FastaFile *__restrict__ files,
                                                         uint8_t *__restrict__ bases, uint32_t *__restrict__ kept_rel,
                                                         uint32_t *__restrict__ kept_len, uint32_t *__restrict__ all_len,
                                                         FastaResult *__restrict__ results)
{
    const FastaFile F = files[blockIdx.x];
    const uint32_t lane = threadIdx.x;
    const uint8_t *tx = text + F.text_off;
    uint8_t *out = bases + F.out_off;
    uint32_t *k_rel = kept_rel + F.table_off, *k_len = kept_len + F.table_off, *a_len = all_len + F.table_off;

    // wave-uniform state
    bool in_header = false;               // the line the next byte belongs to is a header line
    bool open_real = false;               // the open record began with a header (text in front of the first header: a pseudo-record)
    uint32_t open_len = 0;                // bases of the open record so far
    uint32_t open_start = 0;              // its start in the output region (multiple of 32)
    uint32_t open_hdr = 0xFFFFFFFFu;      // text offset of its '>'
    uint32_t n_kept = 0, n_lens = 0, flags = 0, first_hdr = 0xFFFFFFFFu;

    bool last_nl = true;                  // the byte in front of the round is a line end ('\n' in front of the file)
    uint32_t wd[16];                      // the lane's 64 bytes of this round; nx: of the next one (loaded a round ahead)
    load_chunk(wd, tx, (uint64_t)lane * FA_CHUNK, F.text_len);

    for (uint64_t r0 = 0; r0 < F.text_len; r0 += FA_ROUND) {
        const uint64_t c0 = r0 + (uint64_t)lane * FA_CHUNK;
        const uint32_t n_valid = c0 >= F.text_len ? 0u : (F.text_len - c0 < FA_CHUNK ? (uint32_t)(F.text_len - c0) : FA_CHUNK);
        uint32_t nx[16];
        load_chunk(nx, tx, c0 + FA_ROUND, F.text_len);

        // ---- the chunk as three 64-bit masks, bit i = byte i: line ends, '>', bytes below 0x21 (line ends, '\r', blanks and
        // other control bytes: none of them is a base).  Bytes behind the end of the file are line ends.
        uint64_t NL = chunk_mask(wd, [](uint32_t w) { return zero_bytes(w ^ 0x0A0A0A0Au); });
        uint64_t GT = chunk_mask(wd, [](uint32_t w) { return zero_bytes(w ^ 0x3E3E3E3Eu); });
        uint64_t LOW = chunk_mask(wd, [](uint32_t w) { return ~((w | 0x80808080u) - 0x21212121u) & ~w & 0x80808080u; });
        if (n_valid < FA_CHUNK) {
            const uint64_t valid = (1ull << n_valid) - 1ull;
            NL |= ~valid; LOW |= ~valid; GT &= valid;
        }
        const uint32_t up_nl = (uint32_t)__shfl_up((int)(uint32_t)(NL >> 63), 1, 64);
        const bool prev_nl = lane ? up_nl != 0u : last_nl;
        const uint64_t LS = (NL << 1) | (prev_nl ? 1ull : 0ull);        // line starts
        const uint64_t H = LS & GT;                                       // record starts: '>' at a line start

        // ---- the kind of the LAST line start of the chunk decides the state of the lanes behind it
        const bool last_is_hdr = LS != 0ull && ((H >> (63 - __clzll((long long)LS))) & 1ull) != 0ull;
        const unsigned long long m_any = __ballot(LS != 0ull), m_hdrline = __ballot(last_is_hdr);
        const int pk = prev_set_below(m_any, lane);
        bool in_hdr = pk < 0 ? in_header : ((m_hdrline >> pk) & 1ull) != 0ull;     // state at the chunk's first byte

        // ---- header bytes: from a record start (or the chunk start, if it lies inside a header line) up to the next line
        // start.  A carry injected behind every record start runs through the 1-bits of ~LS and stops AT the next line start:
        // the bits the addition flips, moved down by one, are the header bytes.
        const uint64_t P = ~LS, R = P + (H << 1) + (in_hdr ? 1ull : 0ull), flipped = R ^ P;
        const uint64_t INH = (flipped >> 1) | H | (flipped & P & (1ull << 63));
        in_hdr = (INH >> 63) != 0ull;                                    // state behind the chunk's last byte
        const uint64_t BASE = ~INH & ~LOW;
        // anything below 0x21 inside a sequence line that is not a line end: fine if it is a '\r' in front of a line end,
        // otherwise the host parses the file (blanks; control bytes, which the host counts as bases)
        const uint64_t odd = LOW & ~NL & ~INH;
        bool bad = false;
        if (__any(odd != 0ull)) {
            const uint64_t CR = chunk_mask(wd, [](uint32_t w) { return zero_bytes(w ^ 0x0D0D0D0Du); });
            const uint32_t down_nl = (uint32_t)__shfl_down((int)(uint32_t)(NL & 1ull), 1, 64);
            const uint32_t next_first = (uint32_t)__shfl((int)(nx[0] & 0xFFu), 0, 64);      // the next round's first byte
            const bool next_nl = lane == 63u ? next_first == (uint32_t)'\n' : down_nl != 0u;
            const uint64_t nl_next = (NL >> 1) | (next_nl ? 1ull << 63 : 0ull);
            bad = (odd & ~(CR & nl_next)) != 0ull;
        }

        // ---- head = bases in front of the chunk's first record start (they belong to the record open at the chunk start),
        // tail = bases behind its last record start; records in between are complete inside the chunk (and too short to keep)
        const uint32_t n_hdr = (uint32_t)__popcll(H);
        const uint32_t first_hdr_at = n_hdr ? (uint32_t)__ffsll((long long)H) - 1u : 0u, last_hdr_at = n_hdr ? 63u - (uint32_t)__clzll((long long)H) : 0u;
        const uint64_t HEADM = n_hdr ? BASE & ((1ull << first_hdr_at) - 1ull) : BASE;
        const uint64_t TAILM = n_hdr ? BASE & ~((2ull << last_hdr_at) - 1ull) : 0ull;
        const uint32_t head = (uint32_t)__popcll(HEADM), tail = (uint32_t)__popcll(TAILM);
        uint32_t n_tiny = 0;              // non-empty records between two record starts of the chunk
        if (n_hdr > 1u) {
            uint64_t hh = H & (H - 1ull);
            uint32_t from = first_hdr_at;
            while (hh) {
                const uint32_t to = (uint32_t)__ffsll((long long)hh) - 1u;
                if (BASE & ((1ull << to) - 1ull) & ~((2ull << from) - 1ull)) n_tiny++;
                from = to; hh &= hh - 1ull;
            }
        }
        if (__any(bad)) flags |= 1u;
        const unsigned long long m_h = __ballot(n_hdr != 0u);

        // ---- the record open at the lane's chunk start: bases in front of the chunk, where it starts in the output, whether
        // it closes in this round (at the first lane with a record start at or behind this one) and how long it is then
        const uint32_t head_incl = wave_incl_scan_u32(head), head_excl = head_incl - head;
        // padded size of the record that closes at this lane's first record start
        const int opener = prev_set_below(m_h, lane);                  // lane whose last record start opened it, or -1: the carry
        const uint32_t op_tail = (uint32_t)__shfl((int)tail, opener < 0 ? 0 : opener, 64);
        const uint32_t op_hexcl = (uint32_t)__shfl((int)head_excl, opener < 0 ? 0 : opener, 64);
        const uint32_t op_head = (uint32_t)__shfl((int)head, opener < 0 ? 0 : opener, 64);
        const uint32_t before = opener < 0 ? open_len + head_excl : op_tail + head_excl - op_hexcl - op_head;    // bases of the open record in front of this chunk
        const bool op_real = opener < 0 ? open_real : true;
        const uint32_t closed_len = before + head;                      // (meaningful on lanes with a record start)
        const bool closes_kept = n_hdr != 0u && op_real && closed_len >= (uint32_t)ANI_MIN_CONTIG;
        const uint32_t pad = closes_kept ? ((closed_len + 31u) & ~31u) : 0u;
        const uint32_t pad_incl = wave_incl_scan_u32(pad);
        // output start of the record open at this lane's chunk start: the cursor behind everything closed up to its opener
        const uint32_t op_padincl = (uint32_t)__shfl((int)pad_incl, opener < 0 ? 0 : opener, 64);
        const uint32_t rec_start = open_start + (opener < 0 ? 0u : op_padincl);
        // does it close in this round, and is it kept?
        const int closer = next_set_from(m_h, lane);
        const uint32_t cl_len = (uint32_t)__shfl((int)closed_len, closer < 0 ? 0 : closer, 64);
        const bool write_head = closer < 0 ? op_real : (op_real && cl_len >= (uint32_t)ANI_MIN_CONTIG);
        // the record opened by this lane's last record start: starts behind this lane's closing; closes at the next lane with one
        const int closer2 = lane == 63u ? -1 : next_set_from(m_h, lane + 1u);
        const uint32_t c2_hexcl = (uint32_t)__shfl((int)head_excl, closer2 < 0 ? 0 : closer2, 64);
        const uint32_t c2_head = (uint32_t)__shfl((int)head, closer2 < 0 ? 0 : closer2, 64);
        const uint32_t tail_total = closer2 < 0 ? 0u : tail + c2_hexcl - head_excl - head + c2_head;
        const bool write_tail = n_hdr != 0u && (closer2 < 0 || tail_total >= (uint32_t)ANI_MIN_CONTIG);
        const uint32_t tail_start = open_start + pad_incl;

        // ---- tables.  Closed records: their lengths for N50 (if not empty); kept ones into the record table, in order
        {
            const bool emits = n_hdr != 0u && closed_len != 0u;
            const uint32_t n_emit = (emits ? 1u : 0u) + n_tiny;
            const uint32_t e_incl = wave_incl_scan_u32(n_emit), e_at = n_lens + e_incl - n_emit;
            const uint32_t e_tot = (uint32_t)__shfl((int)e_incl, 63, 64);
            if (n_lens + e_tot > F.rec_cap) flags |= 4u;
            else {
                uint32_t at = e_at;
                if (emits) a_len[at++] = closed_len;
                if (n_tiny) {
                    uint64_t hh = H & (H - 1ull);
                    uint32_t from = first_hdr_at;
                    while (hh) {
                        const uint32_t to = (uint32_t)__ffsll((long long)hh) - 1u;
                        const uint32_t cnt = (uint32_t)__popcll(BASE & ((1ull << to) - 1ull) & ~((2ull << from) - 1ull));
                        if (cnt) a_len[at++] = cnt;
                        from = to; hh &= hh - 1ull;
                    }
                }
            }
            n_lens += e_tot;
            const uint32_t k_incl = wave_incl_scan_u32(closes_kept ? 1u : 0u), k_tot = (uint32_t)__shfl((int)k_incl, 63, 64);
            if (n_kept + k_tot > F.rec_cap) flags |= 4u;
            else if (closes_kept) {
                const uint32_t at = n_kept + k_incl - 1u;
                k_rel[at] = rec_start; k_len[at] = closed_len;
            }
            // the header of the first kept record: the text offset of its '>' comes from its opener
            const uint32_t my_last_hdr = (uint32_t)(c0 + last_hdr_at);
            // (the shuffle outside the select: inside it only the lanes with an opener would be active, and a read from an
            //  inactive lane returns 0)
            const uint32_t op_last = (uint32_t)__shfl((int)my_last_hdr, opener < 0 ? 0 : opener, 64);
            const uint32_t op_hdr = opener < 0 ? open_hdr : op_last;
            const unsigned long long m_k = __ballot(closes_kept);
            if (first_hdr == 0xFFFFFFFFu && m_k) {
                const int fl = __ffsll((long long)m_k) - 1;
                first_hdr = (uint32_t)__shfl((int)op_hdr, fl, 64);
            }
            n_kept += k_tot;
            if (closed_len > 0x7FFFFFFFu) flags |= 2u;
        }

        // ---- bases and padding.  The head bases go behind what the open record had in front of the chunk, the tail bases to
        // the start of the record the chunk's last record start opens; byte by byte, the rank of a base among the lane's
        // written ones is its offset (head and tail differ in the base address only)
        {
            // (a region of bound(text length) bytes always has room -- bases <= text bytes, 31 bytes of padding per record of
            //  >= 500 bases: the limits below only keep a wrong table from writing outside the region)
            const uint64_t o_head = (uint64_t)rec_start + before, o_tail = tail_start;
            bool over = false;
            uint64_t Wh = write_head ? HEADM : 0ull, Wt = write_tail ? TAILM : 0ull;
            if (Wh && o_head + head > F.out_cap) { over = true; Wh = 0ull; }
            if (Wt && o_tail + tail > F.out_cap) { over = true; Wt = 0ull; }
            const uint64_t W = Wh | Wt;
            const uint32_t a_head = (uint32_t)o_head, a_tail = (uint32_t)o_tail - (uint32_t)__popcll(Wh);
            const uint32_t split = n_hdr ? last_hdr_at : 64u;
            const uint32_t w_lo = (uint32_t)W, w_hi = (uint32_t)(W >> 32);
            uint32_t k = 0;
#pragma unroll
            for (uint32_t i = 0; i < FA_CHUNK; i++) {
                const uint32_t bit = ((i < 32u ? w_lo : w_hi) >> (i & 31u)) & 1u;
                if (bit) out[(i < split ? a_head : a_tail) + k] = (uint8_t)(wd[i >> 2] >> (8u * (i & 3u)));
                k += bit;
            }
            if (closes_kept) {
                if ((uint64_t)rec_start + pad <= F.out_cap) for (uint32_t q = closed_len; q < pad; q++) out[rec_start + q] = 'A';
                else over = true;
            }
            if (__any(over)) flags |= 8u;
        }

        // ---- carry
        in_header = (bool)__shfl((int)in_hdr, 63, 64);
        last_nl = __shfl((int)(uint32_t)(NL >> 63), 63, 64) != 0;
        const uint32_t head_tot = (uint32_t)__shfl((int)head_incl, 63, 64);
        if (m_h) {
            const int L = 63 - __clzll((long long)m_h);
            const uint32_t l_tail = (uint32_t)__shfl((int)tail, L, 64), l_hexcl = (uint32_t)__shfl((int)head_excl, L, 64);
            const uint32_t l_head = (uint32_t)__shfl((int)head, L, 64);
            open_len = l_tail + head_tot - l_hexcl - l_head;
            open_start += (uint32_t)__shfl((int)pad_incl, 63, 64);
            open_hdr = (uint32_t)__shfl((int)(uint32_t)(c0 + last_hdr_at), L, 64);
            open_real = true;
        } else {
            open_len += head_tot;
        }
        if (open_len > 0x7FFFFFFFu) { flags |= 2u; break; }
#pragma unroll
        for (int q = 0; q < 16; q++) wd[q] = nx[q];
    }

    // ---- the end of the file closes the open record
    if (lane == 0) {
        if (open_len) { if (n_lens < F.rec_cap) a_len[n_lens] = open_len; else flags |= 4u; n_lens++; }
        uint32_t end = open_start;
        if (open_real && open_len >= (uint32_t)ANI_MIN_CONTIG) {
            if (n_kept < F.rec_cap) { k_rel[n_kept] = open_start; k_len[n_kept] = open_len; } else flags |= 4u;
            if (first_hdr == 0xFFFFFFFFu) first_hdr = open_hdr;
            n_kept++;
            const uint32_t pad = (open_len + 31u) & ~31u;
            if ((uint64_t)open_start + pad <= F.out_cap) for (uint32_t k = open_len; k < pad; k++) out[open_start + k] = 'A';
            else flags |= 8u;
            end = open_start + pad;
        }
        FastaResult R;
        R.n_kept = n_kept; R.n_lens = n_lens; R.flags = flags; R.packed_size = end; R.first_hdr = first_hdr;
        R.pad[0] = R.pad[1] = R.pad[2] = 0;
        results[blockIdx.x] = R;
    }
}

void fasta_parse_launch(const uint8_t *d_text, const FastaFile *d_files, uint32_t n_files, uint8_t *d_bases, uint32_t *d_kept_rel,
                        uint32_t *d_kept_len, uint32_t *d_all_len, FastaResult *d_results, hipStream_t st)
{
    if (!n_files) return;
    hipLaunchKernelGGL(fasta_parse_kernel, dim3(n_files), dim3(64), 0, st, d_text, d_files, d_bases, d_kept_rel, d_kept_len, d_all_len, d_results);
    HIPCHECK(hipGetLastError());
}

// =============================================================================================================================
// THE TILED PARSER.  One wavefront per file is as fast as the PCIe copy it follows only while a batch holds a few hundred files:
// a wavefront walks its file 4 KB at a time (106 GB/s over a batch of 170 files, 0.8 GB/s for a single genome).  Here every 4 KB
// TILE of every file is a wavefront of its own, in three kernels:
//   A  fasta_tile_sum_kernel   per tile, WITHOUT knowing what came before it: does the tile contain a line start, and is the last
//      one a header line; the bases in front of its first line start (they count only if the tile begins inside a sequence line);
//      the bases in front of its first record start and behind its last one; the records that begin AND end inside the tile (how
//      many are not empty, how many are kept, their padded size).  Everything else about a tile is fixed by its own bytes.
//   B  fasta_tile_scan_kernel  per file, one wavefront over the tile summaries in order (64 loaded at a time, a few dozen
//      instructions per tile): the state every tile starts in -- inside a header line or not, the record that is open, its length
//      so far and where it starts in the output, how many table entries precede --, the table entries of the records that cross
//      tiles, the end of the file.
//   C  fasta_tile_write_kernel per tile again, now with that state: the bases to their places, the padding, the table entries
//      of the records inside the tile.
// The text is read twice (A and C), the bases are written once; the result is the one-wavefront-per-file kernel's, byte for byte
// (same per-lane masks, same scans inside a tile: a tile is what a round is there).
struct TileSum {
    uint32_t flags;              // 1 has a line start, 2 the last one begins a header line, 4 blanks behind the first line start, 8 blanks in front of it, 16 has a record start
    uint32_t p0;                 // bases in front of the first line start, if the tile begins inside a sequence line
    uint32_t head_det;           // bases from the first line start to the first record start (to the end of the tile if there is none)
    uint32_t tail;               // bases behind the last record start
    uint32_t inner_nonempty, inner_kept, inner_pad;   // records opened and closed inside the tile: with bases; of >= 500; their padded sizes
    uint32_t last_hdr_pos;       // text offset of the tile's last record start
    uint32_t first_inner_kept_hdr;                     // ... of the record start that opens its first kept inner record
    uint32_t pad[3];
};
struct TileCarry {
    uint32_t flags;              // 1 begins inside a header line, 2 the open record began with a header
    uint32_t open_len, open_start;                     // the record open at the tile's first byte: bases so far, start in the output
    uint32_t rid;                // its number among the file's tile-crossing records (kept_flag[rid]); the record open at the tile's end: rid + 1 if the tile has a record start
    uint32_t kept_base, lens_base;                     // table indices of the tile's first inner record
    uint32_t pad[2];
};
#define TILE_BYTES FA_ROUND
#define TILE_STAGE 4608u          // LDS window of a tile's output: 4096 bases, the padding of the records that end in it, alignment

// the file of tile g: last file whose tile_off <= g (files without tiles share an offset with their successor: the LAST of equal offsets has the tiles)
__device__ __forceinline__ uint32_t file_of_tile(const FastaFile *__restrict__ files, uint32_t n_files, uint32_t g)
{
    uint32_t lo = 0, hi = n_files;
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (files[mid].tile_off <= g) lo = mid; else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o, 64);
    return v;
}

// what kernels A and C share: the lane's masks and counts, given the state at the tile's first byte
struct LaneView {
    uint64_t NL, LOW, LS, H, INH, BASE, HEADM, TAILM, prefix;
    uint32_t n_hdr, first_hdr_at, last_hdr_at, head, tail, n_tiny;
    unsigned long long m_any;
    bool in_hdr_out;             // state behind the lane's last byte
};

__device__ __forceinline__ void lane_view(LaneView &V, const uint32_t (&wd)[16], uint32_t n_valid, bool first_prev_nl, bool tile_in_hdr, uint32_t lane)
{
    uint64_t NL = chunk_mask(wd, [](uint32_t w) { return zero_bytes(w ^ 0x0A0A0A0Au); });
    uint64_t GT = chunk_mask(wd, [](uint32_t w) { return zero_bytes(w ^ 0x3E3E3E3Eu); });
    uint64_t LOW = chunk_mask(wd, [](uint32_t w) { return ~((w | 0x80808080u) - 0x21212121u) & ~w & 0x80808080u; });
    if (n_valid < FA_CHUNK) {
        const uint64_t valid = (1ull << n_valid) - 1ull;
        NL |= ~valid; LOW |= ~valid; GT &= valid;
    }
    const uint32_t up_nl = (uint32_t)__shfl_up((int)(uint32_t)(NL >> 63), 1, 64);
    const bool prev_nl = lane ? up_nl != 0u : first_prev_nl;
    const uint64_t LS = (NL << 1) | (prev_nl ? 1ull : 0ull);
    const uint64_t H = LS & GT;
    const bool last_is_hdr = LS != 0ull && ((H >> (63 - __clzll((long long)LS))) & 1ull) != 0ull;
    const unsigned long long m_any = __ballot(LS != 0ull), m_hdrline = __ballot(last_is_hdr);
    const int pk = prev_set_below(m_any, lane);
    const bool in_hdr = pk < 0 ? tile_in_hdr : ((m_hdrline >> pk) & 1ull) != 0ull;
    const uint64_t P = ~LS, R = P + (H << 1) + (in_hdr ? 1ull : 0ull), flipped = R ^ P;
    const uint64_t INH = (flipped >> 1) | H | (flipped & P & (1ull << 63));
    V.NL = NL; V.LOW = LOW; V.LS = LS; V.H = H; V.INH = INH; V.BASE = ~INH & ~LOW; V.m_any = m_any;
    V.in_hdr_out = (INH >> 63) != 0ull;
    // bytes in front of the tile's first line start
    V.prefix = pk < 0 ? (LS ? (LS & (0ull - LS)) - 1ull : ~0ull) : 0ull;
    V.n_hdr = (uint32_t)__popcll(H);
    V.first_hdr_at = V.n_hdr ? (uint32_t)__ffsll((long long)H) - 1u : 0u;
    V.last_hdr_at = V.n_hdr ? 63u - (uint32_t)__clzll((long long)H) : 0u;
    V.HEADM = V.n_hdr ? V.BASE & ((1ull << V.first_hdr_at) - 1ull) : V.BASE;
    V.TAILM = V.n_hdr ? V.BASE & ~((2ull << V.last_hdr_at) - 1ull) : 0ull;
    V.head = (uint32_t)__popcll(V.HEADM); V.tail = (uint32_t)__popcll(V.TAILM);
    V.n_tiny = 0;
    if (V.n_hdr > 1u) {
        uint64_t hh = H & (H - 1ull);
        uint32_t from = V.first_hdr_at;
        while (hh) {
            const uint32_t to = (uint32_t)__ffsll((long long)hh) - 1u;
            if (V.BASE & ((1ull << to) - 1ull) & ~((2ull << from) - 1ull)) V.n_tiny++;
            from = to; hh &= hh - 1ull;
        }
    }
}

// blanks and control bytes outside header lines that are not a '\r' in front of a line end (bit mask over the lane's bytes, header
// lines NOT yet excluded: the caller knows which bytes are header bytes)
__device__ __forceinline__ uint64_t odd_bytes(const LaneView &V, const uint32_t (&wd)[16], bool next_tile_nl, uint32_t lane)
{
    const uint64_t odd = V.LOW & ~V.NL;
    if (!__any(odd != 0ull)) return 0ull;
    const uint64_t CR = chunk_mask(wd, [](uint32_t w) { return zero_bytes(w ^ 0x0D0D0D0Du); });
    const uint32_t down_nl = (uint32_t)__shfl_down((int)(uint32_t)(V.NL & 1ull), 1, 64);
    const bool next_nl = lane == 63u ? next_tile_nl : down_nl != 0u;
    const uint64_t nl_next = (V.NL >> 1) | (next_nl ? 1ull << 63 : 0ull);
    return odd & ~(CR & nl_next);
}

__global__ __launch_bounds__(256) void fasta_tile_sum_kernel(const uint8_t *__restrict__ text, const FastaFile *__restrict__ files, uint32_t n_files,
                                                             uint32_t total_tiles, TileSum *__restrict__ sums)
{
    const uint32_t lane = threadIdx.x & 63u, g = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (g >= total_tiles) return;
    const FastaFile F = files[file_of_tile(files, n_files, g)];
    const uint8_t *tx = text + F.text_off;
    const uint64_t r0 = (uint64_t)(g - F.tile_off) * TILE_BYTES, c0 = r0 + (uint64_t)lane * FA_CHUNK;
    const uint32_t n_valid = c0 >= F.text_len ? 0u : (F.text_len - c0 < FA_CHUNK ? (uint32_t)(F.text_len - c0) : FA_CHUNK);
    uint32_t wd[16];
    load_chunk(wd, tx, c0, F.text_len);
    const bool first_prev_nl = tx[(int64_t)r0 - 1] == '\n';                                      // ('\n' in front of the file)
    const bool next_tile_nl = r0 + TILE_BYTES >= F.text_len || tx[r0 + TILE_BYTES] == '\n';
    LaneView V;
    lane_view(V, wd, n_valid, first_prev_nl, true, lane);       // as if the tile began inside a header line: the bytes in front of its first line start count for nothing
    const uint64_t odd = odd_bytes(V, wd, next_tile_nl, lane);
    const bool bad_det = __any((odd & ~V.INH) != 0ull), bad_prefix = __any((odd & V.prefix) != 0ull);
    const uint32_t p0 = wave_sum_u32((uint32_t)__popcll(~V.LOW & V.prefix));
    const unsigned long long m_h = __ballot(V.n_hdr != 0u);
    const uint32_t head_incl = wave_incl_scan_u32(V.head), head_excl = head_incl - V.head;
    const uint32_t head_tot = (uint32_t)__shfl((int)head_incl, 63, 64);
    // records opened and closed inside the tile: they close at a lane with a record start whose opener is a lane of the tile
    const int opener = prev_set_below(m_h, lane);
    const uint32_t op_tail = (uint32_t)__shfl((int)V.tail, opener < 0 ? 0 : opener, 64);
    const uint32_t op_hexcl = (uint32_t)__shfl((int)head_excl, opener < 0 ? 0 : opener, 64);
    const uint32_t op_head = (uint32_t)__shfl((int)V.head, opener < 0 ? 0 : opener, 64);
    const bool inner = V.n_hdr != 0u && opener >= 0;
    const uint32_t closed_len = inner ? op_tail + head_excl - op_hexcl - op_head + V.head : 0u;
    const bool kept_inner = inner && closed_len >= (uint32_t)ANI_MIN_CONTIG;
    const uint32_t inner_kept = wave_sum_u32(kept_inner ? 1u : 0u);
    const uint32_t inner_pad = wave_sum_u32(kept_inner ? ((closed_len + 31u) & ~31u) : 0u);
    const uint32_t inner_nonempty = wave_sum_u32((inner && closed_len != 0u ? 1u : 0u) + V.n_tiny);
    const uint32_t my_last_hdr = (uint32_t)(c0 + V.last_hdr_at);
    const uint32_t op_last = (uint32_t)__shfl((int)my_last_hdr, opener < 0 ? 0 : opener, 64);
    const unsigned long long m_k = __ballot(kept_inner);
    const uint32_t first_inner_kept_hdr = m_k ? (uint32_t)__shfl((int)op_last, __ffsll((long long)m_k) - 1, 64) : 0xFFFFFFFFu;
    uint32_t head_det = head_tot, tail = 0u, last_hdr_pos = 0xFFFFFFFFu;
    if (m_h) {
        const int FL = __ffsll((long long)m_h) - 1, L = 63 - __clzll((long long)m_h);
        head_det = (uint32_t)__shfl((int)head_incl, FL, 64);
        const uint32_t l_tail = (uint32_t)__shfl((int)V.tail, L, 64), l_hexcl = (uint32_t)__shfl((int)head_excl, L, 64), l_head = (uint32_t)__shfl((int)V.head, L, 64);
        tail = l_tail + head_tot - l_hexcl - l_head;
        last_hdr_pos = (uint32_t)__shfl((int)my_last_hdr, L, 64);
    }
    const bool state_out = (bool)__shfl((int)V.in_hdr_out, 63, 64);
    if (lane == 0) {
        TileSum S;
        S.flags = (V.m_any ? 1u : 0u) | (state_out ? 2u : 0u) | (bad_det ? 4u : 0u) | (bad_prefix ? 8u : 0u) | (m_h ? 16u : 0u);
        S.p0 = p0; S.head_det = head_det; S.tail = tail;
        S.inner_nonempty = inner_nonempty; S.inner_kept = inner_kept; S.inner_pad = inner_pad;
        S.last_hdr_pos = last_hdr_pos; S.first_inner_kept_hdr = first_inner_kept_hdr;
        S.pad[0] = S.pad[1] = S.pad[2] = 0;
        sums[g] = S;
    }
}

__global__ __launch_bounds__(64) void fasta_tile_scan_kernel(const FastaFile *__restrict__ files, const TileSum *__restrict__ sums,
                                                             TileCarry *__restrict__ carries, uint8_t *__restrict__ kept_flag,
                                                             uint8_t *__restrict__ bases, uint32_t *__restrict__ kept_rel, uint32_t *__restrict__ kept_len,
                                                             uint32_t *__restrict__ all_len, FastaResult *__restrict__ results)
{
    const FastaFile F = files[blockIdx.x];
    const uint32_t lane = threadIdx.x, nt = (uint32_t)(((uint64_t)F.text_len + TILE_BYTES - 1u) / TILE_BYTES);
    uint32_t *k_rel = kept_rel + F.table_off, *k_len = kept_len + F.table_off, *a_len = all_len + F.table_off;
    uint8_t *kf = kept_flag + F.tile_off + blockIdx.x;            // this file's records that cross tiles: at most one per tile, and the one open at the end
    uint8_t *out = bases + F.out_off;
    // wave-uniform state (every lane computes the same; lane 0 writes the tables)
    bool in_hdr = false, open_real = false;
    uint32_t open_len = 0, open_start = 0, open_hdr = 0xFFFFFFFFu, n_kept = 0, n_lens = 0, flags = 0, first_hdr = 0xFFFFFFFFu, rid = 0;
    for (uint32_t b0 = 0; b0 < nt; b0 += 64u) {
        const uint32_t nb = nt - b0 < 64u ? nt - b0 : 64u;
        TileSum my;
        if (lane < nb) my = sums[F.tile_off + b0 + lane];
        else { my.flags = 0; my.p0 = my.head_det = my.tail = my.inner_nonempty = my.inner_kept = my.inner_pad = 0; my.last_hdr_pos = my.first_inner_kept_hdr = 0xFFFFFFFFu; }
        TileCarry mine;
        mine.flags = 0; mine.open_len = mine.open_start = mine.rid = mine.kept_base = mine.lens_base = 0; mine.pad[0] = mine.pad[1] = 0;
        for (uint32_t j = 0; j < nb; j++) {
            const uint32_t s_flags = (uint32_t)__shfl((int)my.flags, (int)j, 64), s_p0 = (uint32_t)__shfl((int)my.p0, (int)j, 64);
            const uint32_t s_head = (uint32_t)__shfl((int)my.head_det, (int)j, 64);
            if (lane == j) { mine.flags = (in_hdr ? 1u : 0u) | (open_real ? 2u : 0u); mine.open_len = open_len; mine.open_start = open_start; mine.rid = rid; }
            const uint32_t head = s_head + (in_hdr ? 0u : s_p0);
            if ((s_flags & 4u) || (!in_hdr && (s_flags & 8u))) flags |= 1u;
            if (!(s_flags & 16u)) {
                open_len += head;
            } else {
                const uint32_t s_tail = (uint32_t)__shfl((int)my.tail, (int)j, 64), s_ine = (uint32_t)__shfl((int)my.inner_nonempty, (int)j, 64);
                const uint32_t s_ik = (uint32_t)__shfl((int)my.inner_kept, (int)j, 64), s_ip = (uint32_t)__shfl((int)my.inner_pad, (int)j, 64);
                const uint32_t s_lh = (uint32_t)__shfl((int)my.last_hdr_pos, (int)j, 64), s_fk = (uint32_t)__shfl((int)my.first_inner_kept_hdr, (int)j, 64);
                const uint32_t closed = open_len + head;
                if (closed > 0x7FFFFFFFu) flags |= 2u;
                if (closed) { if (lane == 0 && n_lens < F.rec_cap) a_len[n_lens] = closed; n_lens++; }
                const bool kept = open_real && closed >= (uint32_t)ANI_MIN_CONTIG;
                if (lane == 0) kf[rid] = kept ? 1 : 0;
                if (kept) {
                    if (lane == 0 && n_kept < F.rec_cap) { k_rel[n_kept] = open_start; k_len[n_kept] = closed; }
                    if (first_hdr == 0xFFFFFFFFu) first_hdr = open_hdr;
                    n_kept++;
                    open_start += (closed + 31u) & ~31u;
                }
                if (lane == j) { mine.kept_base = n_kept; mine.lens_base = n_lens; }
                if (first_hdr == 0xFFFFFFFFu && s_ik) first_hdr = s_fk;
                n_kept += s_ik; n_lens += s_ine; open_start += s_ip;
                rid++;
                open_len = s_tail; open_real = true; open_hdr = s_lh;
            }
            if (open_len > 0x7FFFFFFFu) flags |= 2u;
            if (s_flags & 1u) in_hdr = (s_flags & 2u) != 0u;
        }
        if (lane < nb) carries[F.tile_off + b0 + lane] = mine;
    }
    // ---- the end of the file closes the open record
    if (lane == 0) {
        if (open_len) { if (n_lens < F.rec_cap) a_len[n_lens] = open_len; n_lens++; }
        uint32_t end = open_start;
        const bool kept = open_real && open_len >= (uint32_t)ANI_MIN_CONTIG;
        kf[rid] = kept ? 1 : 0;
        if (kept) {
            if (n_kept < F.rec_cap) { k_rel[n_kept] = open_start; k_len[n_kept] = open_len; }
            if (first_hdr == 0xFFFFFFFFu) first_hdr = open_hdr;
            n_kept++;
            const uint32_t pad = (open_len + 31u) & ~31u;
            if ((uint64_t)open_start + pad <= F.out_cap) for (uint32_t k = open_len; k < pad; k++) out[open_start + k] = 'A';
            else flags |= 8u;
            end = open_start + pad;
        }
        if (n_kept > F.rec_cap || n_lens > F.rec_cap) flags |= 4u;
        FastaResult R;
        R.n_kept = n_kept; R.n_lens = n_lens; R.flags = flags; R.packed_size = end; R.first_hdr = first_hdr;
        R.pad[0] = R.pad[1] = R.pad[2] = 0;
        results[blockIdx.x] = R;
    }
}

__global__ __launch_bounds__(256) void fasta_tile_write_kernel(const uint8_t *__restrict__ text, const FastaFile *__restrict__ files, uint32_t n_files,
                                                               uint32_t total_tiles, const TileCarry *__restrict__ carries,
                                                               const uint8_t *__restrict__ kept_flag, uint8_t *__restrict__ bases,
                                                               uint32_t *__restrict__ kept_rel, uint32_t *__restrict__ kept_len,
                                                               uint32_t *__restrict__ all_len, FastaResult *__restrict__ results)
{
    const uint32_t lane = threadIdx.x & 63u, g = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (g >= total_tiles) return;
    const uint32_t fi = file_of_tile(files, n_files, g);
    const FastaFile F = files[fi];
    const TileCarry T = carries[g];
    const uint8_t *tx = text + F.text_off;
    uint8_t *out = bases + F.out_off;
    uint32_t *k_rel = kept_rel + F.table_off, *k_len = kept_len + F.table_off, *a_len = all_len + F.table_off;
    const uint8_t *kf = kept_flag + F.tile_off + fi;
    const uint64_t r0 = (uint64_t)(g - F.tile_off) * TILE_BYTES, c0 = r0 + (uint64_t)lane * FA_CHUNK;
    const uint32_t n_valid = c0 >= F.text_len ? 0u : (F.text_len - c0 < FA_CHUNK ? (uint32_t)(F.text_len - c0) : FA_CHUNK);
    uint32_t wd[16];
    load_chunk(wd, tx, c0, F.text_len);
    const bool first_prev_nl = tx[(int64_t)r0 - 1] == '\n';
    LaneView V;
    lane_view(V, wd, n_valid, first_prev_nl, (T.flags & 1u) != 0u, lane);
    const bool open_real = (T.flags & 2u) != 0u;
    const uint32_t open_len = T.open_len, open_start = T.open_start;
    const uint32_t n_hdr = V.n_hdr, head = V.head, tail = V.tail;
    const unsigned long long m_h = __ballot(n_hdr != 0u);
    // ---- exactly the round of fasta_parse_kernel from here, with the carry read instead of kept
    const uint32_t head_incl = wave_incl_scan_u32(head), head_excl = head_incl - head;
    const int opener = prev_set_below(m_h, lane);
    const uint32_t op_tail = (uint32_t)__shfl((int)tail, opener < 0 ? 0 : opener, 64);
    const uint32_t op_hexcl = (uint32_t)__shfl((int)head_excl, opener < 0 ? 0 : opener, 64);
    const uint32_t op_head = (uint32_t)__shfl((int)head, opener < 0 ? 0 : opener, 64);
    const uint32_t before = opener < 0 ? open_len + head_excl : op_tail + head_excl - op_hexcl - op_head;
    const bool op_real = opener < 0 ? open_real : true;
    const uint32_t closed_len = before + head;
    const bool closes_kept = n_hdr != 0u && op_real && closed_len >= (uint32_t)ANI_MIN_CONTIG;
    const uint32_t pad = closes_kept ? ((closed_len + 31u) & ~31u) : 0u;
    const uint32_t pad_incl = wave_incl_scan_u32(pad);
    const uint32_t op_padincl = (uint32_t)__shfl((int)pad_incl, opener < 0 ? 0 : opener, 64);
    const uint32_t rec_start = open_start + (opener < 0 ? 0u : op_padincl);
    const int closer = next_set_from(m_h, lane);
    const uint32_t cl_len = (uint32_t)__shfl((int)closed_len, closer < 0 ? 0 : closer, 64);
    // a record that does not close inside the tile: the scan knows whether it is kept (the one open at the tile's start, or the
    // one its last record start opens)
    const bool keep_open = kf[T.rid] != 0, keep_tail = m_h ? kf[T.rid + 1u] != 0 : keep_open;
    const bool write_head = closer < 0 ? (opener < 0 ? keep_open : keep_tail) : (op_real && cl_len >= (uint32_t)ANI_MIN_CONTIG);
    const int closer2 = lane == 63u ? -1 : next_set_from(m_h, lane + 1u);
    const uint32_t c2_hexcl = (uint32_t)__shfl((int)head_excl, closer2 < 0 ? 0 : closer2, 64);
    const uint32_t c2_head = (uint32_t)__shfl((int)head, closer2 < 0 ? 0 : closer2, 64);
    const uint32_t tail_total = closer2 < 0 ? 0u : tail + c2_hexcl - head_excl - head + c2_head;
    const bool write_tail = n_hdr != 0u && (closer2 < 0 ? keep_tail : tail_total >= (uint32_t)ANI_MIN_CONTIG);
    const uint32_t tail_start = open_start + pad_incl;
    // ---- table entries of the records inside the tile (the one that closes at the tile's first record start is the scan's)
    {
        const bool inner = n_hdr != 0u && opener >= 0;
        const uint32_t n_emit = (inner && closed_len != 0u ? 1u : 0u) + V.n_tiny;
        const uint32_t e_incl = wave_incl_scan_u32(n_emit);
        uint32_t at = T.lens_base + e_incl - n_emit;
        if (inner && closed_len != 0u) { if (at < F.rec_cap) a_len[at] = closed_len; at++; }
        if (V.n_tiny) {
            uint64_t hh = V.H & (V.H - 1ull);
            uint32_t from = V.first_hdr_at;
            while (hh) {
                const uint32_t to = (uint32_t)__ffsll((long long)hh) - 1u;
                const uint32_t cnt = (uint32_t)__popcll(V.BASE & ((1ull << to) - 1ull) & ~((2ull << from) - 1ull));
                if (cnt) { if (at < F.rec_cap) a_len[at] = cnt; at++; }
                from = to; hh &= hh - 1ull;
            }
        }
        const bool kept_inner = inner && closes_kept;
        const uint32_t k_incl = wave_incl_scan_u32(kept_inner ? 1u : 0u);
        if (kept_inner) {
            const uint32_t ka = T.kept_base + k_incl - 1u;
            if (ka < F.rec_cap) { k_rel[ka] = rec_start; k_len[ka] = closed_len; }
        }
    }
    // ---- bases and padding.  What a tile writes is ONE contiguous range of the output (the end of the record open at its start,
    // padding, the kept records inside it, the beginning of the record open at its end; records that are dropped leave no hole:
    // the next one takes their place).  The lanes put their bytes into the wavefront's LDS window and the window goes out in
    // 16-byte stores -- byte stores from 64 lanes to 64 places took 2.0 ms per 450 MB batch, four times the rest of the parser.
    // The partial units at both ends are stored byte by byte: their other bytes are a neighbouring tile's.
    {
        __shared__ __attribute__((aligned(16))) uint8_t stage_all[4][TILE_STAGE];
        uint8_t *stage = stage_all[threadIdx.x >> 6];
        const uint64_t o_head = (uint64_t)rec_start + before, o_tail = tail_start;
        bool over = false;
        uint64_t Wh = write_head ? V.HEADM : 0ull, Wt = write_tail ? V.TAILM : 0ull;
        if (Wh && o_head + head > F.out_cap) { over = true; Wh = 0ull; }
        if (Wt && o_tail + tail > F.out_cap) { over = true; Wt = 0ull; }
        bool pads = closes_kept;
        if (pads && (uint64_t)rec_start + pad > F.out_cap) { over = true; pads = false; }
        const uint32_t nh = (uint32_t)__popcll(Wh), nt = (uint32_t)__popcll(Wt);
        uint32_t lo = 0xFFFFFFFFu, hi = 0u;
        if (nh) { lo = (uint32_t)o_head; hi = (uint32_t)o_head + nh; }
        if (nt) { lo = lo < (uint32_t)o_tail ? lo : (uint32_t)o_tail; hi = hi > (uint32_t)o_tail + nt ? hi : (uint32_t)o_tail + nt; }
        if (pads && pad > closed_len) { lo = lo < rec_start + closed_len ? lo : rec_start + closed_len; hi = hi > rec_start + pad ? hi : rec_start + pad; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const uint32_t l2 = (uint32_t)__shfl_xor((int)lo, o, 64), h2 = (uint32_t)__shfl_xor((int)hi, o, 64);
            lo = lo < l2 ? lo : l2; hi = hi > h2 ? hi : h2;
        }
        const uint32_t w0 = lo & ~15u;
        const bool staged = lo < hi && hi - w0 <= TILE_STAGE;          // (always, unless the tables are wrong: then byte stores)
        const uint64_t W = Wh | Wt;
        const uint32_t a_head = (uint32_t)o_head, a_tail = (uint32_t)o_tail - nh;
        const uint32_t split = n_hdr ? V.last_hdr_at : 64u;
        const uint32_t w_lo = (uint32_t)W, w_hi = (uint32_t)(W >> 32);
        if (!staged) {
            uint32_t k = 0;
#pragma unroll
            for (uint32_t i = 0; i < FA_CHUNK; i++) {
                const uint32_t bit = ((i < 32u ? w_lo : w_hi) >> (i & 31u)) & 1u;
                if (bit) out[(i < split ? a_head : a_tail) + k] = (uint8_t)(wd[i >> 2] >> (8u * (i & 3u)));
                k += bit;
            }
            if (pads) for (uint32_t q = closed_len; q < pad; q++) out[rec_start + q] = 'A';
        } else {
            const uint32_t s_head = a_head - w0, s_tail = a_tail - w0;     // (window offsets; a lane without a head or a tail never uses its one)
            uint32_t k = 0;
#pragma unroll
            for (uint32_t i = 0; i < FA_CHUNK; i++) {
                const uint32_t bit = ((i < 32u ? w_lo : w_hi) >> (i & 31u)) & 1u;
                if (bit) stage[(i < split ? s_head : s_tail) + k] = (uint8_t)(wd[i >> 2] >> (8u * (i & 3u)));
                k += bit;
            }
            if (pads) for (uint32_t q = closed_len; q < pad; q++) stage[rec_start + q - w0] = 'A';
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (uint32_t u = lane * 16u; w0 + u < hi; u += 64u * 16u) {
                const uint32_t b = w0 + u;
                if (b >= lo && b + 16u <= hi) {
                    *reinterpret_cast<uint4 *>(out + b) = *reinterpret_cast<const uint4 *>(stage + u);
                } else {
                    for (uint32_t j = 0; j < 16u; j++)
                        if (b + j >= lo && b + j < hi) out[b + j] = stage[u + j];
                }
            }
        }
        if (__any(over) && lane == 0) atomicOr(&results[fi].flags, 8u);
    }
}

size_t fasta_tiles_work_bytes(uint32_t total_tiles, uint32_t n_files)
{
    return ((size_t)total_tiles + 1u) * (sizeof(TileSum) + sizeof(TileCarry)) + (((size_t)total_tiles + n_files + 1u + 15u) & ~(size_t)15u);
}

void fasta_parse_tiles_launch(const uint8_t *d_text, const FastaFile *d_files, uint32_t n_files, uint32_t total_tiles, void *d_work, uint8_t *d_bases,
                              uint32_t *d_kept_rel, uint32_t *d_kept_len, uint32_t *d_all_len, FastaResult *d_results, hipStream_t st)
{
    if (!n_files) return;
    TileSum *sums = static_cast<TileSum *>(d_work);
    TileCarry *carries = reinterpret_cast<TileCarry *>(sums + total_tiles + 1u);
    uint8_t *kept_flag = reinterpret_cast<uint8_t *>(carries + total_tiles + 1u);
    if (total_tiles)
        hipLaunchKernelGGL(fasta_tile_sum_kernel, dim3((total_tiles + 3u) / 4u), dim3(256), 0, st, d_text, d_files, n_files, total_tiles, sums);
    hipLaunchKernelGGL(fasta_tile_scan_kernel, dim3(n_files), dim3(64), 0, st, d_files, sums, carries, kept_flag, d_bases, d_kept_rel, d_kept_len,
                       d_all_len, d_results);
    if (total_tiles)
        hipLaunchKernelGGL(fasta_tile_write_kernel, dim3((total_tiles + 3u) / 4u), dim3(256), 0, st, d_text, d_files, n_files, total_tiles, carries,
                           kept_flag, d_bases, d_kept_rel, d_kept_len, d_all_len, d_results);
    HIPCHECK(hipGetLastError());
}
