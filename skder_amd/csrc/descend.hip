// descend.hip -- DESCENDANTS OF REAL ASSEMBLIES generated on the device (bench.py, tests): the benchmark workloads with real
// genome structure (the reference's published run is > 20,000 genomes of ONE genus, /root/reference/README.md:27) need tens of
// thousands of related genomes; numpy builds one in ~50 ms of CPU, this builds 20,000 of them (43 Gb) in well under a second.
//
// Model (counter-based, like synth.h: everything is a pure function of the descendant's seed and a position, so the two
// passes -- lengths, then bases -- agree without any state between them).  A descendant keeps its parent's records (contigs);
//   structural  at most one event per record, at most three per genome (host-chosen, skder_descendant_t::ev): an inversion, a
//               translocation inside the record, or a deletion of 0.5 - 20 kb -- the record after the event is the VIRTUAL record;
//   short indels  at every virtual position with probability indel_ppm / 10^6 an event starts: an insertion of l random bases
//               behind the position or a deletion of l positions from it on, l geometric with mean 2.5 (capped at 15);
//   substitutions at every virtual position with probability sub_ppm / 10^6 (A/C/G/T only; other letters stay).
// Two kernels over tiles of 4096 virtual positions: descend_count_kernel (output bases per tile -> a device-wide scan gives
// the record lengths and every tile's place) and descend_fill_kernel (the bases, into the 32-byte-aligned batch layout the
// sketch kernel reads).
#include "device_utils.h"
#include "engine.h"
#include "synth.h"

#define DSC_TILE 4096u
#define DSC_PER 16u            // positions per thread
#define DSC_MAXL 15u           // longest short indel

struct DscTile {
    uint32_t desc;             // descendant index in the call
    uint32_t rec;              // record index inside the call (all descendants' records concatenated)
    uint32_t v0, nv;           // virtual positions [v0, v0 + nv) of the record
    uint32_t first_tile;       // the record's first tile
    uint32_t rec_local;        // the record's index inside its genome: with the descendant's seed, the hash coordinates (independent of the call's batch)
    uint32_t pad[2];
};
struct DscRec {
    uint64_t anc_off;          // the parent record's place in the ancestors' buffer
    uint32_t anc_len, vlen;    // parent length; virtual length (after the structural event)
    uint32_t ev_type, ev_s, ev_n, ev_b;    // structural event inside this record (ev_type 3: none)
};

// the short-indel event that starts at hash coordinate g: 0 = none, else length (1..15) | 16 for a deletion
__device__ __forceinline__ uint32_t dsc_event(uint64_t seed, uint64_t g, uint32_t indel_ppm)
{
    const uint64_t e = synth_h(seed ^ 0x1DE1ULL, g);
    if ((uint32_t)(e % 1000000u) >= indel_ppm) return 0u;
    const uint32_t r = (uint32_t)(e >> 44) & 0xFFFFu;
    // P(l > k) = 0.6^k: thresholds 65536 * 0.6^k
    uint32_t l = 1u;
    l += r < 39322u; l += r < 23593u; l += r < 14156u; l += r < 8493u; l += r < 5096u; l += r < 3058u; l += r < 1835u;
    l += r < 1101u; l += r < 660u; l += r < 396u; l += r < 238u; l += r < 143u; l += r < 86u; l += r < 51u;
    return l | (((uint32_t)(e >> 40) & 1u) << 4);
}

// virtual position v of a record -> position in the parent record; rc: read the complement
__device__ __forceinline__ uint32_t dsc_map(const DscRec &R, uint32_t v, bool &rc)
{
    rc = false;
    if (R.ev_type == 0u) {                 // inversion of [s, s + n)
        if (v >= R.ev_s && v < R.ev_s + R.ev_n) { rc = true; return R.ev_s + (R.ev_s + R.ev_n - 1u - v); }
        return v;
    }
    if (R.ev_type == 2u) return v < R.ev_s ? v : v + R.ev_n;                 // deletion
    if (R.ev_type == 1u) {                 // [s, s + n) moved to position b of the record without it
        if (v >= R.ev_b && v < R.ev_b + R.ev_n) return R.ev_s + (v - R.ev_b);
        const uint32_t r = v < R.ev_b ? v : v - R.ev_n;
        return r < R.ev_s ? r : r + R.ev_n;
    }
    return v;
}

__device__ __forceinline__ uint8_t dsc_complement(uint8_t c)
{
    switch (c) {
    case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A';
    case 'a': return 't'; case 'c': return 'g'; case 'g': return 'c'; case 't': return 'a';
    default: return c;
    }
}

// the tile's events into LDS (with the DSC_MAXL positions in front of it), then per position: deleted or not, bases inserted behind it
template <bool FILL>
__global__ __launch_bounds__(256) void descend_kernel(const uint8_t *__restrict__ anc, const DscTile *__restrict__ tiles, const DscRec *__restrict__ recs,
                                                      const skder_descendant_t *__restrict__ desc, uint32_t *__restrict__ tile_count,
                                                      const uint32_t *__restrict__ tile_off, uint8_t *__restrict__ out,
                                                      const uint64_t *__restrict__ out_rec_off)
{
    __shared__ uint8_t evb[DSC_TILE + DSC_MAXL + 1];
    __shared__ uint32_t ws[4];
    const DscTile T = tiles[blockIdx.x];
    const DscRec R = recs[T.rec];
    const skder_descendant_t D = desc[T.desc];
    const uint64_t gbase = (uint64_t)T.rec_local << 32;        // hash coordinates: (record of the genome, virtual position)
    for (uint32_t k = threadIdx.x; k < T.nv + DSC_MAXL; k += 256u) {
        const int64_t v = (int64_t)T.v0 + (int64_t)k - (int64_t)DSC_MAXL;
        evb[k] = v >= 0 ? (uint8_t)dsc_event(D.seed, gbase + (uint64_t)v, D.indel_ppm) : (uint8_t)0;
    }
    __syncthreads();
    const uint32_t p0 = threadIdx.x * DSC_PER;
    uint32_t cnt = 0, keep = 0, insl[DSC_PER];
#pragma unroll
    for (uint32_t u = 0; u < DSC_PER; u++) {
        const uint32_t p = p0 + u;
        insl[u] = 0;
        if (p >= T.nv) continue;
        bool deleted = false;
#pragma unroll
        for (uint32_t k = 0; k < DSC_MAXL; k++) {
            const uint32_t e = evb[p + DSC_MAXL - k];
            deleted |= (e & 16u) && (e & 15u) > k;
        }
        const uint32_t e0 = evb[p + DSC_MAXL];
        const uint32_t ins = (e0 && !(e0 & 16u)) ? (e0 & 15u) : 0u;
        keep |= deleted ? 0u : 1u << u;
        insl[u] = ins;
        cnt += (deleted ? 0u : 1u) + ins;
    }
    uint32_t total;
    const uint32_t ex = block_excl_scan_256(cnt, ws, total);
    if (!FILL) {
        if (threadIdx.x == 0) tile_count[blockIdx.x] = total;
        return;
    }
    const uint64_t place = out_rec_off[T.rec];
    if (place == ~0ull) return;                 // a record the caller dropped (shorter than 500 bases after the edits); uniform per block
    uint8_t *o = out + place + (tile_off[blockIdx.x] - tile_off[T.first_tile]) + ex;
    const uint8_t *a = anc + R.anc_off;
#pragma unroll
    for (uint32_t u = 0; u < DSC_PER; u++) {
        const uint32_t p = p0 + u;
        if (p >= T.nv) break;
        const uint32_t v = T.v0 + p;
        if ((keep >> u) & 1u) {
            bool rc;
            const uint32_t src = dsc_map(R, v, rc);
            uint8_t c = a[src];
            if (rc) c = dsc_complement(c);
            const uint64_t h = synth_h(D.seed ^ 0x5AB5ULL, gbase + v);
            if ((uint32_t)(h % 1000000u) < D.sub_ppm) {
                const uint8_t up = c & 0xDFu;
                const int code = up == 'A' ? 0 : up == 'C' ? 1 : up == 'G' ? 2 : up == 'T' ? 3 : -1;
                if (code >= 0) c = "ACGT"[(code + 1 + (int)((h >> 32) % 3u)) & 3];
            }
            *o++ = c;
        }
        for (uint32_t j = 0; j < insl[u]; j++) *o++ = "ACGT"[synth_h(D.seed ^ 0x1257ULL, (gbase + v) * 16u + j) & 3u];
    }
}

namespace {
struct DscPlan {
    std::vector<DscTile> tiles;
    std::vector<DscRec> recs;
    std::vector<uint32_t> rec_first_tile;     // + one entry behind the last record
};

DscPlan make_plan(const skder_batch_t *anc, const skder_descendant_t *desc, uint32_t n_desc)
{
    DscPlan P;
    for (uint32_t d = 0; d < n_desc; d++) {
        const skder_descendant_t &D = desc[d];
        if (D.parent >= anc->n_genomes) throw SkError("descend: parent index beyond the ancestors' batch");
        if (D.n_events > 3u) throw SkError("descend: at most three structural events per descendant");
        const uint32_t r0 = anc->genome_rec_begin[D.parent], r1 = anc->genome_rec_begin[D.parent + 1];
        for (uint32_t r = r0; r < r1; r++) {
            DscRec R;
            R.anc_off = anc->rec_off[r]; R.anc_len = anc->rec_len[r]; R.vlen = R.anc_len;
            R.ev_type = 3u; R.ev_s = R.ev_n = R.ev_b = 0;
            for (uint32_t e = 0; e < D.n_events; e++) {
                if (D.ev[e].rec != r - r0) continue;
                if (R.ev_type != 3u) throw SkError("descend: two structural events in one record");
                const uint64_t s = D.ev[e].s, n = D.ev[e].n;
                if (D.ev[e].type > 2u || n == 0 || s + n > R.anc_len) throw SkError("descend: structural event outside its record");
                if (D.ev[e].type == 1u && (uint64_t)D.ev[e].b > R.anc_len - n) throw SkError("descend: translocation target outside its record");
                R.ev_type = D.ev[e].type; R.ev_s = D.ev[e].s; R.ev_n = D.ev[e].n; R.ev_b = D.ev[e].b;
                if (R.ev_type == 2u) R.vlen = R.anc_len - R.ev_n;
            }
            const uint32_t rec = (uint32_t)P.recs.size();
            const uint32_t first = (uint32_t)P.tiles.size();
            P.rec_first_tile.push_back(first);
            P.recs.push_back(R);
            for (uint32_t v0 = 0; v0 < R.vlen; v0 += DSC_TILE) {
                DscTile t;
                memset(&t, 0, sizeof t);
                t.desc = d; t.rec = rec; t.rec_local = r - r0; t.v0 = v0; t.nv = R.vlen - v0 < DSC_TILE ? R.vlen - v0 : DSC_TILE; t.first_tile = first;
                P.tiles.push_back(t);
            }
        }
    }
    P.rec_first_tile.push_back((uint32_t)P.tiles.size());
    if (P.tiles.size() >= 0x7FFFFFFFull) throw SkError("descend: too many tiles in one call");
    return P;
}

// tiles / records / descendants on the device, the per-tile counts and their exclusive scan (one entry more than tiles)
struct DscDevice {
    DevBuf<DscTile> tiles;
    DevBuf<DscRec> recs;
    DevBuf<skder_descendant_t> desc;
    DevBuf<uint32_t> count, off;
    ScanWorkspace ws;
};

void run_counts(skder_ctx *ctx, const uint8_t *d_anc, const DscPlan &P, const skder_descendant_t *desc, uint32_t n_desc, DscDevice &V)
{
    hipStream_t st = ctx->stream;
    const size_t nt = P.tiles.size();
    V.tiles.resize(nt + 1, st); V.recs.resize(P.recs.size() + 1, st); V.desc.resize(n_desc + 1, st);
    V.count.resize(nt + 1, st); V.off.resize(nt + 1, st);
    HIPCHECK(hipMemcpyAsync(V.tiles.p, P.tiles.data(), nt * sizeof(DscTile), hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(V.recs.p, P.recs.data(), P.recs.size() * sizeof(DscRec), hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(V.desc.p, desc, n_desc * sizeof(skder_descendant_t), hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemsetAsync(V.count.p + nt, 0, 4, st));
    if (nt) hipLaunchKernelGGL(descend_kernel<false>, dim3((unsigned)nt), dim3(256), 0, st, d_anc, V.tiles.p, V.recs.p, V.desc.p, V.count.p, nullptr, nullptr, nullptr);
    HIPCHECK(hipGetLastError());
    exclusive_scan_u32(V.count.p, V.off.p, nt + 1, V.ws, st);
}
}   // namespace

#define DSC_CATCH(CTX)                                                                              \
    catch (const std::exception &e) { if (CTX) (CTX)->last_error = e.what(); return 2; }

extern "C" int skder_amd_descend_lengths(skder_ctx_t *ctx, const uint8_t *d_anc, const skder_batch_t *anc, const skder_descendant_t *desc,
                                         uint32_t n_desc, uint32_t *rec_len_out, uint32_t n_rec_out)
{
    if (!ctx || !d_anc || !anc || !desc || !rec_len_out) return 1;
    try {
        HIPCHECK(hipSetDevice(ctx->device));
        const DscPlan P = make_plan(anc, desc, n_desc);
        if (P.recs.size() != n_rec_out) throw SkError("descend_lengths: rec_len_out must hold one entry per record of every descendant's parent");
        DscDevice V;
        run_counts(ctx, d_anc, P, desc, n_desc, V);
        std::vector<uint32_t> off(P.tiles.size() + 1);
        HIPCHECK(hipMemcpyAsync(off.data(), V.off.p, off.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHECK(hipStreamSynchronize(ctx->stream));
        if (!off.empty() && (uint64_t)off.back() > 0xF0000000ull) throw SkError("descend: more than 4 Gb in one call (split the descendants into batches)");
        for (size_t r = 0; r < P.recs.size(); r++) rec_len_out[r] = off[P.rec_first_tile[r + 1]] - off[P.rec_first_tile[r]];
        return 0;
    }
    DSC_CATCH(ctx)
}

extern "C" int skder_amd_descend_fill(skder_ctx_t *ctx, const uint8_t *d_anc, const skder_batch_t *anc, const skder_descendant_t *desc,
                                      uint32_t n_desc, uint8_t *d_out, const uint64_t *rec_out_off, uint32_t n_rec_out)
{
    if (!ctx || !d_anc || !anc || !desc || !d_out || !rec_out_off) return 1;
    try {
        HIPCHECK(hipSetDevice(ctx->device));
        const DscPlan P = make_plan(anc, desc, n_desc);
        if (P.recs.size() != n_rec_out) throw SkError("descend_fill: rec_out_off must hold one entry per record of every descendant's parent");
        DscDevice V;
        run_counts(ctx, d_anc, P, desc, n_desc, V);
        DevBuf<uint64_t> d_off;
        d_off.resize(n_rec_out + 1, ctx->stream);
        HIPCHECK(hipMemcpyAsync(d_off.p, rec_out_off, (size_t)n_rec_out * 8, hipMemcpyHostToDevice, ctx->stream));
        if (!P.tiles.empty())
            hipLaunchKernelGGL(descend_kernel<true>, dim3((unsigned)P.tiles.size()), dim3(256), 0, ctx->stream, d_anc, V.tiles.p, V.recs.p, V.desc.p, nullptr,
                               V.off.p, d_out, d_off.p);
        HIPCHECK(hipGetLastError());
        HIPCHECK(hipStreamSynchronize(ctx->stream));
        return 0;
    }
    DSC_CATCH(ctx)
}
