// sketch.hip -- FracMinHash sketching of genomes resident in HBM (gfx950).
//
// Replaces the sketching stage of `skani triangle|sketch|search|dist` as spawned by
// /root/reference/src/skDER/skder.py:16-26,103,119 (skani itself is an external binary; the
// algorithm restated here is pinned by oracle/ani_oracle.c sketch_contig()).
//
// Data flow (all on one stream):
//   bases (ASCII, 1 B/base, records 32-B aligned)
//     -> sketch_tiles_kernel   one workgroup per 8192-position tile: 16-B coalesced loads, 2-bit
//                              packing into LDS, each thread rolls 32 consecutive positions of the
//                              15-mer (u32) and 21-mer (u64) forward/reverse registers, hashes both
//                              canonical k-mers, keeps hash < 2^64/c; ordered compaction through a
//                              workgroup scan into fixed-capacity per-tile slots
//     -> scan of per-tile counts, gather_* kernels -> position-ordered seed arrays per genome
//     -> marker_sort_kernel    per genome: bitonic sort in LDS + dedup -> sorted unique markers
// HBM traffic per base: 1 B read (+ 32/8192 halo) ; per seed 12 B written twice, 8 B read once.
#include <algorithm>

#include "common.h"
#include "device_utils.h"
#include "engine.h"
#include "sketch_body.h"

// ---------------------------------------------------------------------------------------------
// device helpers

// 4 ASCII bases in a dword -> 4 two-bit codes in the low byte (A=0 C=1 G=2 T=3, case-insensitive,
// every other byte = 0), first base in the lowest bits.
__device__ __forceinline__ uint32_t pack4(uint32_t x)
{
    uint32_t u = x & 0xDFDFDFDFu;                       // upper-case
    uint32_t c = ((u >> 1) ^ (u >> 2)) & 0x03030303u;   // A0 C1 G2 T3 (N happens to give 0 too)
    // bytes equal to 'C','G','T' (exact zero-byte test, no borrow artefacts)
    uint32_t vc = u ^ 0x43434343u, vg = u ^ 0x47474747u, vt = u ^ 0x54545454u;
    uint32_t nzc = (((vc & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | vc) & 0x80808080u;
    uint32_t nzg = (((vg & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | vg) & 0x80808080u;
    uint32_t nzt = (((vt & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | vt) & 0x80808080u;
    uint32_t valid = (~(nzc & nzg & nzt)) & 0x80808080u;   // 0x80 where the byte is C, G or T
    c &= (valid >> 7) * 3u;
    uint32_t t = c | (c >> 6);
    return (t & 0xFu) | ((t >> 12) & 0xF0u);
}

__device__ __forceinline__ uint32_t pack16(uint4 v)
{
    return pack4(v.x) | (pack4(v.y) << 8) | (pack4(v.z) << 16) | (pack4(v.w) << 24);
}

// reverse complement of a 15-mer held as 30 bits, newest base in the low pair
__device__ __forceinline__ uint32_t revcomp15(uint32_t f)
{
    uint32_t x = __brev(f);
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    return (~(x >> 2)) & SK_SEED_MASK;
}
__device__ __forceinline__ uint64_t revcomp21(uint64_t f)
{
    uint64_t x = __brevll(f);
    x = ((x >> 1) & 0x5555555555555555ULL) | ((x & 0x5555555555555555ULL) << 1);
    return (~(x >> 22)) & SK_MARK_MASK;
}

// ---------------------------------------------------------------------------------------------
// HOT KERNEL
//
// The kernel is bound by integer issue, not by HBM (1.07 B of traffic per base; profiles/): per position it rolls the
// forward / reverse 21-mer registers, derives both 15-mers from them, takes the two canonical k-mers and runs mm_hash64 on
// each.  The per-thread body lives in sketch_body.h, one option bit per instruction-selection choice;
// profiles/calib/sketch_body_bench.hip times the combinations on the device (profiles/round3_sketch_body.json): the compiler's
// own selection costs 107 ns per position and wavefront, the body the kernel runs (SK_BODY_DEFAULT) 93:
//   * the 15-mers are NOT rolled separately: the forward one is the low 30 bits of the forward 21-mer, the reverse one the
//     top 30 bits of the reverse 21-mer (-5.5 ns);
//   * threshold test: v_cmp_gt_u64 against the constant in an SGPR pair, then v_addc_co_u32 mask, mask, mask -- the carry
//     shifts into the mask; position j of the lane's 32 ends up at bit 31 - j, one v_bfrev_b32 per mask at the end (-3.5 ns
//     against the compiler's compare / v_cndmask / v_or3 chain);
//   * x * 21 as two v_lshl_add_u64 instead of two v_mad_u64_u32 and two register moves (-0.7 ns);
//   * the canonical 21-mer through ONE v_min_f64: 42-bit integers are denormal doubles, which order like the integers they
//     are (f64 denormals are never flushed on this target); the compiler's v_cmp_lt_u64 + two v_cndmask cost 7.6 ns, this 2.8.
// What did NOT pay, measured in the same table: v_mul_lo_u32, v_mad_u32_u24 and v_lshl_add_u32 for the high words of the
// multiplications (each several times the price of a v_mad_u64_u32, which is cheap here -- the compiler's pair of multiply-adds
// with two register moves in between is hard to beat), and both hashes written out as one assembly stream on fixed
// registers (strictly in order: slower than what the compiler's scheduler makes of the same arithmetic).
// (-DSK_BODY_DEFAULT=0 builds the compiler-selected body: A/B measurements, the parity reference of the others.)

#define PACKED_WORDS ((SKDER_TILE + 32) / 16)   // 514

template <int VARIANT>
__global__ __launch_bounds__(SK_THREADS) void sketch_tiles_kernel(
    const uint8_t *__restrict__ bases, const TileDesc *__restrict__ tiles,
    uint32_t *__restrict__ slot_kmer, uint32_t *__restrict__ slot_gpos, uint64_t *__restrict__ slot_mark,
    uint32_t *__restrict__ tile_ns, uint32_t *__restrict__ tile_nm, uint32_t *__restrict__ flags)
{
    __shared__ uint32_t packed[PACKED_WORDS + 6];
    __shared__ uint32_t wsum[SK_THREADS / 64];
    __shared__ uint32_t out_kmer[SK_SLOT_SEEDS];
    __shared__ uint32_t out_gpos[SK_SLOT_SEEDS];
    __shared__ uint64_t out_mark[SK_SLOT_MARKS];

    const uint32_t tid = threadIdx.x;
    const TileDesc td = tiles[blockIdx.x];
    const uint8_t *src = bases + td.base_off - 32;   // 32-B aligned; 32 bases of left halo

    for (uint32_t w = tid; w < PACKED_WORDS; w += SK_THREADS) {
        uint4 v = *reinterpret_cast<const uint4 *>(src + 16ull * w);
        packed[w] = pack16(v);
    }
    if (tid < 6) packed[PACKED_WORDS + tid] = 0;
    __syncthreads();

    // this thread's 64-base window: tile positions [32*tid - 32, 32*tid + 32)
    const uint32_t w0 = packed[2 * tid], w1 = packed[2 * tid + 1];
    const uint32_t w2 = packed[2 * tid + 2], w3 = packed[2 * tid + 3];

    uint32_t smask, mmask;
    const uint32_t p0 = tid * SK_POS_PER_THREAD;
    sketch_body<VARIANT>(w0, w1, w2, w3, smask, mmask);
    // validity: inside the record, and at least marker_k-1 bases in front (ani_oracle.c sketch_contig)
    {
        uint32_t valid = 0xFFFFFFFFu;
        if (p0 + 32 > td.npos) valid = (p0 >= td.npos) ? 0u : (0xFFFFFFFFu >> (32 - (td.npos - p0)));
        uint32_t i0 = td.pos0 + p0;   // record position of j = 0
        if (i0 < ANI_MARKER_K - 1) {
            uint32_t skip = (ANI_MARKER_K - 1) - i0;
            valid &= (skip >= 32) ? 0u : (0xFFFFFFFFu << skip);
        }
        smask &= valid;
        mmask &= valid;
    }

    // ordered compaction: workgroup exclusive scan of (seed count | marker count << 16)
    uint32_t cnt = __popc(smask) | (__popc(mmask) << 16);
    uint32_t total;
    uint32_t excl = block_excl_scan_256(cnt, wsum, total);
    uint32_t so = excl & 0xFFFFu, mo = excl >> 16;
    const uint32_t ns = total & 0xFFFFu, nm = total >> 16;

    while (smask) {
        int j = __ffs(smask) - 1;
        smask &= smask - 1;
        // 15-mer ending at window base 32+j: window bases [18+j, 32+j]
        uint32_t n0 = 18 + j;
        uint32_t wi = 2 * tid + (n0 >> 4), sh = 2 * (n0 & 15);
        uint64_t two = ((uint64_t)packed[wi + 1] << 32) | packed[wi];
        uint32_t f = (uint32_t)(two >> sh) & SK_SEED_MASK;
        // window order: older bases in lower bits; the rolling register keeps the NEWEST base lowest
        f = __brev(f);
        f = ((f >> 1) & 0x55555555u) | ((f & 0x55555555u) << 1);
        f >>= 2;                                  // now newest base in the low pair, as fs
        uint32_t r = revcomp15(f);
        uint32_t fwd = f < r;
        if (so < SK_SLOT_SEEDS) {
            out_kmer[so] = (fwd ? f : r) | (fwd ? SK_FWD_BIT : 0u);
            out_gpos[so] = td.gpos0 + p0 + j;
        }
        so++;
    }
    while (mmask) {
        int j = __ffs(mmask) - 1;
        mmask &= mmask - 1;
        uint32_t n0 = 12 + j;                     // 21-mer = window bases [12+j, 32+j]
        uint32_t wi = 2 * tid + (n0 >> 4), sh = 2 * (n0 & 15);
        uint64_t lo = ((uint64_t)packed[wi + 1] << 32) | packed[wi];
        uint64_t v = lo >> sh;
        if (sh > 22) v |= (uint64_t)packed[wi + 2] << (64 - sh);
        v &= SK_MARK_MASK;
        // reverse the base order (oldest-lowest -> newest-lowest)
        uint64_t f = __brevll(v);
        f = ((f >> 1) & 0x5555555555555555ULL) | ((f & 0x5555555555555555ULL) << 1);
        f >>= 22;
        uint64_t r = revcomp21(f);
        if (mo < SK_SLOT_MARKS) out_mark[mo] = f < r ? f : r;
        mo++;
    }
    __syncthreads();
    if (tid == 0) {
        tile_ns[blockIdx.x] = ns;
        tile_nm[blockIdx.x] = nm;
        if (ns > SK_SLOT_SEEDS || nm > SK_SLOT_MARKS) atomicOr(&flags[0], 1u);
    }
    const uint64_t sbase = (uint64_t)blockIdx.x * SK_SLOT_SEEDS;
    for (uint32_t t = tid; t < ns && t < SK_SLOT_SEEDS; t += SK_THREADS) {
        slot_kmer[sbase + t] = out_kmer[t];
        slot_gpos[sbase + t] = out_gpos[t];
    }
    const uint64_t mbase = (uint64_t)blockIdx.x * SK_SLOT_MARKS;
    for (uint32_t t = tid; t < nm && t < SK_SLOT_MARKS; t += SK_THREADS) slot_mark[mbase + t] = out_mark[t];
}

// ---------------------------------------------------------------------------------------------
// gather tile slots into contiguous arrays

__global__ __launch_bounds__(64) void gather_seeds_kernel(
    const TileDesc *__restrict__ tiles, const uint32_t *__restrict__ tile_ns, const uint32_t *__restrict__ tile_soff,
    const uint32_t *__restrict__ slot_kmer, const uint32_t *__restrict__ slot_gpos, uint64_t dst_base,
    uint32_t *__restrict__ seed_kmer, uint32_t *__restrict__ seed_gpos, uint32_t *__restrict__ seed_ctg)
{
    const uint32_t t = blockIdx.x;
    const uint32_t n = tile_ns[t];
    const uint64_t dst = dst_base + tile_soff[t];
    const uint64_t src = (uint64_t)t * SK_SLOT_SEEDS;
    const uint32_t ctg = tiles[t].ctg;
    for (uint32_t i = threadIdx.x; i < n; i += 64) {
        seed_kmer[dst + i] = slot_kmer[src + i];
        seed_gpos[dst + i] = slot_gpos[src + i];
        seed_ctg[dst + i] = ctg;
    }
}

__global__ __launch_bounds__(64) void gather_marks_kernel(
    const uint32_t *__restrict__ tile_nm, const uint32_t *__restrict__ tile_moff,
    const uint64_t *__restrict__ slot_mark, uint64_t *__restrict__ raw_marks)
{
    const uint32_t t = blockIdx.x;
    const uint32_t n = tile_nm[t];
    const uint64_t dst = tile_moff[t];
    const uint64_t src = (uint64_t)t * SK_SLOT_MARKS;
    for (uint32_t i = threadIdx.x; i < n; i += 64) raw_marks[dst + i] = slot_mark[src + i];
}

// per genome: sort the raw markers (bitonic, LDS) and drop duplicates.
// raw[g_off[g] .. g_off[g+1]) -> sorted unique at the same offset; count to n_unique[g].
#define MARK_SORT_CAP 16384
__global__ __launch_bounds__(256) void marker_sort_kernel(uint64_t *__restrict__ raw, const uint32_t *__restrict__ g_off,
                                                          uint32_t *__restrict__ n_unique, uint32_t *__restrict__ flags)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint64_t *key = reinterpret_cast<uint64_t *>(smem_raw);
    __shared__ uint32_t wsum[4];
    const uint32_t g = blockIdx.x;
    const uint32_t lo = g_off[g], n = g_off[g + 1] - lo;
    if (n > MARK_SORT_CAP) return;     // genomes beyond ~16 Mb: sorted in global memory (sort_big_genome_markers)
    uint32_t m = 1;
    while (m < n) m <<= 1;
    for (uint32_t i = threadIdx.x; i < m; i += 256) key[i] = i < n ? raw[lo + i] : ~0ULL;
    __syncthreads();
    for (uint32_t k = 2; k <= m; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = threadIdx.x; t < m / 2; t += 256) {     // one compare-exchange per thread and trip
                const uint32_t i = ((t & ~(j - 1u)) << 1) | (t & (j - 1u)), ixj = i | j;
                const uint64_t a = key[i], b = key[ixj];
                const bool up = (i & k) == 0;
                if ((a > b) == up) { key[i] = b; key[ixj] = a; }
            }
            __syncthreads();
        }
    }
    // ordered dedup: block scan over "is first of its run" in strips of 256
    uint32_t running = 0;
    for (uint32_t base = 0; base < n; base += 256) {
        uint32_t i = base + threadIdx.x;
        uint32_t keep = (i < n) && (i == 0 || key[i] != key[i - 1]);
        uint32_t total;
        uint32_t ex = block_excl_scan_256(keep, wsum, total);
        if (keep) raw[lo + running + ex] = key[i];
        running += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) n_unique[g] = running;
}

// Genomes with more raw markers than marker_sort_kernel's LDS holds (> ~16 Mb): the same bitonic network,
// one launch per compare-exchange step over a padded copy in global memory, then the ordered dedup through
// a device scan.  Rare (large eukaryotic genomes), so launch count is not a concern.
__global__ __launch_bounds__(256) void marks_pad_copy_kernel(const uint64_t *__restrict__ raw, uint32_t lo, uint32_t n, uint32_t m,
                                                             uint64_t *__restrict__ key)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < m) key[i] = i < n ? raw[lo + i] : ~0ULL;
}

__global__ __launch_bounds__(256) void bitonic_step_kernel(uint64_t *__restrict__ key, uint32_t m, uint32_t k, uint32_t j)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= m / 2) return;
    const uint32_t i = ((t & ~(j - 1u)) << 1) | (t & (j - 1u)), ixj = i | j;
    const uint64_t a = key[i], b = key[ixj];
    const bool up = (i & k) == 0;
    if ((a > b) == up) { key[i] = b; key[ixj] = a; }
}

__global__ __launch_bounds__(256) void marks_keep_kernel(const uint64_t *__restrict__ key, uint32_t n, uint32_t *__restrict__ keep)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i <= n) keep[i] = (i < n) && (i == 0 || key[i] != key[i - 1]);
}

__global__ __launch_bounds__(256) void marks_scatter_kernel(const uint64_t *__restrict__ key, const uint32_t *__restrict__ excl, uint32_t n,
                                                            uint64_t *__restrict__ raw, uint32_t lo, uint32_t *__restrict__ n_unique_g)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n && excl[i + 1] != excl[i]) raw[lo + excl[i]] = key[i];
    if (i == 0) *n_unique_g = excl[n];
}

static void sort_big_genome_markers(uint64_t *raw, uint32_t lo, uint32_t n, uint32_t *n_unique_g, ScanWorkspace &ws, hipStream_t st)
{
    uint32_t m = 1;
    while (m < n) m <<= 1;
    DevBuf<uint64_t> key;
    DevBuf<uint32_t> keep, excl;
    key.resize(m, st); keep.resize(n + 1, st); excl.resize(n + 1, st);
    hipLaunchKernelGGL(marks_pad_copy_kernel, dim3((m + 255) / 256), dim3(256), 0, st, raw, lo, n, m, key.p);
    for (uint32_t k = 2; k <= m; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1)
            hipLaunchKernelGGL(bitonic_step_kernel, dim3((m / 2 + 255) / 256), dim3(256), 0, st, key.p, m, k, j);
    hipLaunchKernelGGL(marks_keep_kernel, dim3((n + 256) / 256), dim3(256), 0, st, key.p, n, keep.p);
    exclusive_scan_u32(keep.p, excl.p, n + 1, ws, st);
    hipLaunchKernelGGL(marks_scatter_kernel, dim3((n + 255) / 256), dim3(256), 0, st, key.p, excl.p, n, raw, lo, n_unique_g);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipStreamSynchronize(st));     // the temporaries go back to the pool
}

// compact sorted-unique marker runs (still at their raw offsets) to their final offsets
__global__ __launch_bounds__(256) void marker_compact_kernel(const uint64_t *__restrict__ raw, const uint32_t *__restrict__ g_off,
                                                             const uint32_t *__restrict__ n_unique,
                                                             const uint32_t *__restrict__ dst_off, uint64_t dst_base,
                                                             uint64_t *__restrict__ out)
{
    const uint32_t g = blockIdx.x;
    const uint32_t lo = g_off[g], n = n_unique[g];
    const uint64_t d = dst_base + dst_off[g];
    for (uint32_t i = threadIdx.x; i < n; i += 256) out[d + i] = raw[lo + i];
}

// ---------------------------------------------------------------------------------------------
// host side

// per-record table (host-built, small) from which the tile descriptors are expanded on the device
struct RecDesc {
    uint64_t base_off;    // offset of the record in d_bases
    uint32_t len, genome, ctg, gpos0;
    uint32_t tile_base;   // index of the record's first tile
    uint32_t pad;
};

__global__ __launch_bounds__(256) void make_tiles_kernel(const RecDesc *__restrict__ recs, uint32_t nrec, uint32_t ntiles,
                                                         TileDesc *__restrict__ tiles)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= ntiles) return;
    uint32_t lo = 0, hi = nrec;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (recs[mid].tile_base <= t) lo = mid; else hi = mid;
    }
    const RecDesc r = recs[lo];
    const uint32_t p = (t - r.tile_base) * SKDER_TILE;
    TileDesc d;
    d.base_off = r.base_off + p;
    d.genome = r.genome; d.ctg = r.ctg; d.pos0 = p;
    d.npos = r.len - p < SKDER_TILE ? r.len - p : SKDER_TILE;
    d.gpos0 = r.gpos0 + p;
    d.pad = 0;
    tiles[t] = d;
}

// dst[i] = src[idx[i]]: the per-genome entries of the per-tile offset tables (the host only needs those)
__global__ __launch_bounds__(256) void pick_kernel(const uint32_t *__restrict__ src, const uint32_t *__restrict__ idx, uint32_t n,
                                                   uint32_t *__restrict__ dst)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}

static uint32_t build_records(const skder_batch_t *b, std::vector<RecDesc> &recs, std::vector<uint32_t> &genome_tile_begin,
                              std::vector<uint32_t> &rec_goff /* per genome nrec+1 */, std::vector<uint64_t> &genome_len)
{
    genome_tile_begin.assign(b->n_genomes + 1, 0);
    recs.reserve(b->n_records);
    rec_goff.reserve((size_t)b->n_records + b->n_genomes);
    genome_len.reserve(b->n_genomes);
    uint64_t nt = 0;
    for (uint32_t g = 0; g < b->n_genomes; g++) {
        genome_tile_begin[g] = (uint32_t)nt;
        uint32_t gpos = 0;
        uint32_t r0 = b->genome_rec_begin[g], r1 = b->genome_rec_begin[g + 1];
        for (uint32_t r = r0; r < r1; r++) {
            uint32_t len = b->rec_len[r];
            if (b->rec_off[r] % 32) throw SkError("record offsets must be multiples of 32");
            if ((uint64_t)gpos + len >= 0x7F000000ull) throw SkError("genome longer than 2.1e9 bases");
            rec_goff.push_back(gpos);
            RecDesc d;
            d.base_off = b->rec_off[r]; d.len = len; d.genome = g; d.ctg = r - r0; d.gpos0 = gpos;
            d.tile_base = (uint32_t)nt; d.pad = 0;
            recs.push_back(d);
            nt += (len + SKDER_TILE - 1) / SKDER_TILE;
            if (nt >= 0xFFFF0000ull) throw SkError("too many tiles in one batch");
            gpos += len;
        }
        rec_goff.push_back(gpos);
        genome_len.push_back(gpos);
    }
    genome_tile_begin[b->n_genomes] = (uint32_t)nt;
    return (uint32_t)nt;
}

void sketch_batch_impl(skder_sketches *s, const uint8_t *d_bases, const skder_batch_t *b)
{
    skder_ctx *ctx = s->ctx;
    hipStream_t st = ctx->stream;
    if (s->indexed || s->index_pending) throw SkError("sketch set already indexed; cannot append");
    if (b->n_genomes == 0) return;
    std::vector<RecDesc> recs;
    std::vector<uint32_t> gtb, rec_goff;
    std::vector<uint64_t> glen;
    uint32_t nt = build_records(b, recs, gtb, rec_goff, glen);

    DevBuf<TileDesc> d_tiles;
    DevBuf<RecDesc> d_recs;
    DevBuf<uint32_t> slot_kmer, slot_gpos, tile_ns, tile_nm, tile_soff, tile_moff;
    DevBuf<uint64_t> slot_mark;
    d_tiles.resize(nt, st);
    d_recs.resize(recs.size(), st);
    HIPCHECK(hipMemcpyAsync(d_recs.p, recs.data(), recs.size() * sizeof(RecDesc), hipMemcpyHostToDevice, st));
    if (nt)
        hipLaunchKernelGGL(make_tiles_kernel, dim3((nt + 255) / 256), dim3(256), 0, st, d_recs.p, (uint32_t)recs.size(), nt, d_tiles.p);
    ScanWorkspace ws;
    std::vector<uint32_t> h_soff, h_moff, g_soff, g_moff;     // per tile (refinement only) / per genome boundary
    DevBuf<uint32_t> d_gtb, d_gs, d_gm;
    for (int attempt = 0;; attempt++) {
        slot_kmer.resize((size_t)nt * SK_SLOT_SEEDS, st);
        slot_gpos.resize((size_t)nt * SK_SLOT_SEEDS, st);
        slot_mark.resize((size_t)nt * SK_SLOT_MARKS, st);
        tile_ns.resize(nt + 1, st); tile_nm.resize(nt + 1, st);
        tile_soff.resize(nt + 1, st); tile_moff.resize(nt + 1, st);
        HIPCHECK(hipMemsetAsync(ctx->d_flags, 0, 64, st));
        HIPCHECK(hipMemsetAsync(tile_ns.p + nt, 0, 4, st));
        HIPCHECK(hipMemsetAsync(tile_nm.p + nt, 0, 4, st));

        HIPCHECK(hipEventRecord(ctx->ev[0], st));
        if (nt) {
            hipLaunchKernelGGL(sketch_tiles_kernel<SK_BODY_DEFAULT>, dim3(nt), dim3(SK_THREADS), 0, st, d_bases, d_tiles.p, slot_kmer.p,
                                   slot_gpos.p, slot_mark.p, tile_ns.p, tile_nm.p, ctx->d_flags);
        }
        HIPCHECK(hipEventRecord(ctx->ev[1], st));

        // offsets
        exclusive_scan_u32(tile_ns.p, tile_soff.p, nt + 1, ws, st);
        exclusive_scan_u32(tile_nm.p, tile_moff.p, nt + 1, ws, st);
        // the host needs the offsets at genome boundaries only: pick them on the device (the full per-tile
        // tables are fetched only when tiles have to be refined)
        const uint32_t ng1 = b->n_genomes + 1;
        d_gtb.resize(ng1, st); d_gs.resize(ng1, st); d_gm.resize(ng1, st);
        HIPCHECK(hipMemcpyAsync(d_gtb.p, gtb.data(), ng1 * 4ull, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(pick_kernel, dim3((ng1 + 255) / 256), dim3(256), 0, st, tile_soff.p, d_gtb.p, ng1, d_gs.p);
        hipLaunchKernelGGL(pick_kernel, dim3((ng1 + 255) / 256), dim3(256), 0, st, tile_moff.p, d_gtb.p, ng1, d_gm.p);
        g_soff.resize(ng1); g_moff.resize(ng1);
        uint32_t h_flags = 0;
        HIPCHECK(hipMemcpyAsync(g_soff.data(), d_gs.p, ng1 * 4ull, hipMemcpyDeviceToHost, st));
        HIPCHECK(hipMemcpyAsync(g_moff.data(), d_gm.p, ng1 * 4ull, hipMemcpyDeviceToHost, st));
        HIPCHECK(hipMemcpyAsync(&h_flags, ctx->d_flags, 4, hipMemcpyDeviceToHost, st));
        HIPCHECK(hipStreamSynchronize(st));
        if (!(h_flags & 1u)) break;
        h_soff.resize(nt + 1); h_moff.resize(nt + 1);
        HIPCHECK(hipMemcpy(h_soff.data(), tile_soff.p, (nt + 1) * 4ull, hipMemcpyDeviceToHost));
        HIPCHECK(hipMemcpy(h_moff.data(), tile_moff.p, (nt + 1) * 4ull, hipMemcpyDeviceToHost));
        if (attempt) throw SkError("sketch tile slot overflow after the tiles were refined (internal error)");
        // A tile holds more than 512 seeds or 128 markers in its 8192 positions (low-complexity sequence whose
        // few distinct k-mers happen to be sampled).  Such tiles are cut into 64 pieces of 128 positions --
        // no piece can exceed either capacity -- and the batch is sketched again over the refined tile list.
        std::vector<TileDesc> ht(nt), refined;
        HIPCHECK(hipMemcpy(ht.data(), d_tiles.p, (size_t)nt * sizeof(TileDesc), hipMemcpyDeviceToHost));
        refined.reserve(nt + 1024);
        for (uint32_t t = 0; t < nt; t++) {
            const uint32_t ns = h_soff[t + 1] - h_soff[t], nm = h_moff[t + 1] - h_moff[t];
            if (ns <= SK_SLOT_SEEDS && nm <= SK_SLOT_MARKS) { refined.push_back(ht[t]); continue; }
            for (uint32_t p = 0; p < ht[t].npos; p += 128) {
                TileDesc d = ht[t];
                d.base_off += p; d.pos0 += p; d.gpos0 += p;
                d.npos = ht[t].npos - p < 128 ? ht[t].npos - p : 128;
                refined.push_back(d);
            }
        }
        if (refined.size() >= 0xFFFF0000ull) throw SkError("too many tiles in one batch");
        nt = (uint32_t)refined.size();
        // first tile of every genome in the refined list (tiles stay in genome order)
        for (uint32_t g = 0, t = 0; g <= b->n_genomes; g++) {
            while (t < nt && refined[t].genome < g) t++;
            gtb[g] = t;
        }
        d_tiles.resize(nt, st);
        HIPCHECK(hipMemcpy(d_tiles.p, refined.data(), (size_t)nt * sizeof(TileDesc), hipMemcpyHostToDevice));
    }
    const uint64_t add_seeds = g_soff[b->n_genomes], add_raw_marks = g_moff[b->n_genomes];

    // seeds
    const uint64_t seed_base = s->seed_kmer.n;
    s->seed_kmer.resize(seed_base + add_seeds, st);
    s->seed_gpos.reserve(seed_base + add_seeds + 32, s->seed_gpos.n, st);   // chain_fast_kernel reads whole 64-B lines
    s->seed_gpos.resize(seed_base + add_seeds, st);
    s->seed_ctg.resize(seed_base + add_seeds, st);
    if (nt)
        hipLaunchKernelGGL(gather_seeds_kernel, dim3(nt), dim3(64), 0, st, d_tiles.p, tile_ns.p, tile_soff.p, slot_kmer.p,
                           slot_gpos.p, seed_base, s->seed_kmer.p, s->seed_gpos.p, s->seed_ctg.p);
    // markers: raw -> sorted unique per genome -> appended
    DevBuf<uint64_t> raw_marks;
    raw_marks.resize(add_raw_marks + 1, st);
    if (nt)
        hipLaunchKernelGGL(gather_marks_kernel, dim3(nt), dim3(64), 0, st, tile_nm.p, tile_moff.p, slot_mark.p, raw_marks.p);
    std::vector<uint32_t> h_goff(b->n_genomes + 1);
    for (uint32_t g = 0; g <= b->n_genomes; g++) h_goff[g] = g_moff[g];
    DevBuf<uint32_t> d_goff, d_nuniq, d_uoff;
    d_goff.resize(b->n_genomes + 1, st);
    d_nuniq.resize(b->n_genomes + 1, st);
    d_uoff.resize(b->n_genomes + 1, st);
    HIPCHECK(hipMemcpyAsync(d_goff.p, h_goff.data(), (b->n_genomes + 1) * 4, hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemsetAsync(d_nuniq.p, 0, (b->n_genomes + 1) * 4, st));
    // LDS for the largest genome of the batch only (32 KB for 3 Mb genomes): several workgroups per CU
    uint32_t max_raw = 1024;
    for (uint32_t g = 0; g < b->n_genomes; g++) max_raw = std::max(max_raw, h_goff[g + 1] - h_goff[g]);
    uint32_t sort_cap = 1024;
    while (sort_cap < max_raw && sort_cap < MARK_SORT_CAP) sort_cap <<= 1;
    // (every call: the attribute belongs to the device, and a process may use several)
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(marker_sort_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, MARK_SORT_CAP * 8);
    hipLaunchKernelGGL(marker_sort_kernel, dim3(b->n_genomes), dim3(256), sort_cap * 8, st, raw_marks.p, d_goff.p,
                       d_nuniq.p, ctx->d_flags);
    for (uint32_t g = 0; g < b->n_genomes; g++)
        if (h_goff[g + 1] - h_goff[g] > MARK_SORT_CAP)
            sort_big_genome_markers(raw_marks.p, h_goff[g], h_goff[g + 1] - h_goff[g], d_nuniq.p + g, ws, st);
    exclusive_scan_u32(d_nuniq.p, d_uoff.p, b->n_genomes + 1, ws, st);
    std::vector<uint32_t> h_uoff(b->n_genomes + 1);
    uint32_t h_flags = 0;
    HIPCHECK(hipMemcpyAsync(h_uoff.data(), d_uoff.p, (b->n_genomes + 1) * 4, hipMemcpyDeviceToHost, st));
    HIPCHECK(hipMemcpyAsync(&h_flags, ctx->d_flags, 4, hipMemcpyDeviceToHost, st));
    HIPCHECK(hipStreamSynchronize(st));
    const uint64_t mark_base = s->markers.n;
    s->markers.resize(mark_base + h_uoff[b->n_genomes], st);
    hipLaunchKernelGGL(marker_compact_kernel, dim3(b->n_genomes), dim3(256), 0, st, raw_marks.p, d_goff.p, d_nuniq.p,
                       d_uoff.p, mark_base, s->markers.p);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipEventRecord(ctx->ev[2], st));
    HIPCHECK(hipStreamSynchronize(st));
    float ms0 = 0, ms1 = 0;
    HIPCHECK(hipEventElapsedTime(&ms0, ctx->ev[0], ctx->ev[1]));
    HIPCHECK(hipEventElapsedTime(&ms1, ctx->ev[1], ctx->ev[2]));
    ctx->timing[0] = ms0;
    ctx->timing[1] = ms1;

    // host metadata
    for (uint32_t g = 0; g < b->n_genomes; g++) {
        s->h_seed_off.push_back(seed_base + g_soff[g + 1]);
        s->h_marker_off.push_back(mark_base + h_uoff[g + 1]);
        s->h_genome_len.push_back(glen[g]);
        s->h_genome_nrec.push_back(b->genome_rec_begin[g + 1] - b->genome_rec_begin[g]);
    }
    s->h_rec_goff.insert(s->h_rec_goff.end(), rec_goff.begin(), rec_goff.end());
    s->n_genomes += b->n_genomes;
}
