// fasta.hip -- FASTA text -> device layout, ON the device (gfx950).
//
// The ingest of the drop-in entry points (host_io.hip) used to scan every file on the host: find the lines, drop headers and
// line ends, copy the bases of the records of >= 500 bases into the 32-byte-aligned layout sketch_tiles_kernel reads, count
// record lengths for N50 (util.n50_calc, /root/reference/src/skDER/util.py:686-724).  Here the host only READS (or inflates)
// the file into a pinned buffer; the text goes over PCIe as it is and this kernel does the rest at HBM speed:
//
//   one WAVEFRONT per file, 4 KB of text per round (64 bytes per lane), state carried from round to round in wave-uniform
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
