// screen.hip -- marker containment screen (ani_oracle.c oracle_screen) for all pairs at once.
//
// Instead of N^2/2 pairwise merges (8*(m_i+m_j) bytes each) the reference-side markers go into
// ONE global open-addressing table marker -> list of genomes (an inverted index; 8 B + 4 B per
// marker occurrence), and every row i walks the lists of its own markers, counting shared markers
// per partner genome with LDS atomics.  The pass/fail decision per pair is a bit in a row bitmap;
// a scan over the row popcounts turns the bitmaps into an ordered (i, j) pair list.
#include "device_utils.h"
#include "engine.h"
#include "screen.h"

#define MK_EMPTY 0xFFFFFFFFFFFFFFFFULL
#define SCREEN_JTILE 8192
#define SCREEN_U 4             // markers per wave and trip of screen_rows_kernel
#define SCREEN_THREADS 1024    // 16 waves per row: the kernel is a chain of dependent loads per marker, and a CU's LDS holds
#define SCREEN_WAVES (SCREEN_THREADS / 64u)   // few rows' counters -- more waves per row put more loads in flight

__device__ __forceinline__ uint64_t mix64(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

__global__ void table_clear_kernel(uint64_t *keys, uint32_t *cnt, uint64_t ts)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ts; i += (uint64_t)gridDim.x * blockDim.x) {
        keys[i] = MK_EMPTY;
        cnt[i] = 0;
    }
}

// one workgroup per reference genome: claim a slot for every marker, count occurrences
__global__ __launch_bounds__(256) void table_insert_kernel(const GenomeMeta *__restrict__ meta, const uint64_t *__restrict__ markers,
                                                           unsigned long long *keys, uint32_t *cnt, uint64_t mask,
                                                           uint32_t *__restrict__ slot_of, uint32_t *__restrict__ pos_of)
{
    const GenomeMeta m = meta[blockIdx.x];
    for (uint32_t e = threadIdx.x; e < m.n_markers; e += 256) {
        const uint64_t key = markers[m.marker_off + e];
        uint64_t slot = mix64(key) & mask;
        for (;;) {
            // a slot goes from empty to its key once and stays: a plain load that already shows the key (about 49 of 50
            // occurrences in a set of related genomes) saves the compare-and-swap; a stale "empty" just takes it
            unsigned long long old = __builtin_nontemporal_load(&keys[slot]);
            if (old == MK_EMPTY) old = atomicCAS(&keys[slot], (unsigned long long)MK_EMPTY, (unsigned long long)key);
            if (old == MK_EMPTY || old == key) break;
            slot = (slot + 1) & mask;
        }
        // the count before this occurrence is its place in the marker's genome list: table_fill_kernel needs no atomic of its own
        pos_of[m.marker_off + e] = atomicAdd(&cnt[slot], 1u);
        slot_of[m.marker_off + e] = (uint32_t)slot;
    }
}

__global__ __launch_bounds__(256) void table_fill_kernel(const GenomeMeta *__restrict__ meta, const uint32_t *__restrict__ slot_of,
                                                         const uint32_t *__restrict__ loff, const uint32_t *__restrict__ pos_of,
                                                         uint32_t *__restrict__ list)
{
    const GenomeMeta m = meta[blockIdx.x];
    for (uint32_t e = threadIdx.x; e < m.n_markers; e += 256) {
        const uint32_t slot = slot_of[m.marker_off + e];
        list[loff[slot] + pos_of[m.marker_off + e]] = blockIdx.x;
    }
}

// one workgroup per row: count shared markers with every partner genome of the current j-tile in
// LDS, decide, and write the row's pass bits.  triangle != 0: only partners j > row are counted.
// q_slot_of: slot of each of the row genome's markers in the reference table, or 0xFFFFFFFF.
__global__ __launch_bounds__(SCREEN_THREADS) void screen_rows_kernel(
    const GenomeMeta *__restrict__ qmeta, const uint32_t *__restrict__ q_slot_of, const GenomeMeta *__restrict__ rmeta,
    uint32_t n_ref, const uint32_t *__restrict__ loff, const uint32_t *__restrict__ list,
    const uint32_t *__restrict__ rows, int triangle, double cutoff, int screen_on,
    unsigned long long *__restrict__ pass_bits, uint32_t words_per_row, uint32_t *__restrict__ row_count)
{
    __shared__ uint32_t cnt[SCREEN_JTILE];
    __shared__ uint32_t s_rowcnt;
    const uint32_t row = rows[blockIdx.x];
    const GenomeMeta q = qmeta[row];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid == 0) s_rowcnt = 0;
    for (uint32_t j0 = 0; j0 < n_ref; j0 += SCREEN_JTILE) {
        const uint32_t j1 = j0 + SCREEN_JTILE < n_ref ? j0 + SCREEN_JTILE : n_ref;
        for (uint32_t j = tid; j < SCREEN_JTILE; j += SCREEN_THREADS) cnt[j] = 0;
        __syncthreads();
        if (!(triangle && j1 <= row + 1)) {
            // one wave per marker, SCREEN_U markers per trip: their slots, then their list bounds, then the first
            // 128 entries of every list are requested together (three dependent loads per marker otherwise)
            for (uint32_t e0 = wave; e0 < q.n_markers; e0 += SCREEN_WAVES * SCREEN_U) {
                uint32_t slot[SCREEN_U], lo[SCREEN_U], hi[SCREEN_U], ja[SCREEN_U], jb[SCREEN_U];
#pragma unroll
                for (int u = 0; u < SCREEN_U; u++) {
                    const uint32_t e = e0 + SCREEN_WAVES * u;
                    slot[u] = e < q.n_markers ? q_slot_of[q.marker_off + e] : 0xFFFFFFFFu;
                }
#pragma unroll
                for (int u = 0; u < SCREEN_U; u++) {
                    const bool ok = slot[u] != 0xFFFFFFFFu;
                    lo[u] = ok ? loff[slot[u]] : 0u;
                    hi[u] = ok ? loff[slot[u] + 1] : 0u;
                }
#pragma unroll
                for (int u = 0; u < SCREEN_U; u++) {
                    const uint32_t k = lo[u] + lane;
                    ja[u] = k < hi[u] ? list[k] : 0xFFFFFFFFu;
                    jb[u] = k + 64u < hi[u] ? list[k + 64u] : 0xFFFFFFFFu;
                }
#pragma unroll
                for (int u = 0; u < SCREEN_U; u++) {
                    const uint32_t j = ja[u], j2 = jb[u];
                    if (j >= j0 && j < j1 && (!triangle || j > row)) atomicAdd(&cnt[j - j0], 1u);
                    if (j2 >= j0 && j2 < j1 && (!triangle || j2 > row)) atomicAdd(&cnt[j2 - j0], 1u);
                    for (uint32_t k = lo[u] + 128u + lane; k < hi[u]; k += 64) {
                        const uint32_t j3 = list[k];
                        if (j3 >= j0 && j3 < j1 && (!triangle || j3 > row)) atomicAdd(&cnt[j3 - j0], 1u);
                    }
                }
            }
        }
        __syncthreads();
        // decisions, 64 partners per wave-iteration -> one 64-bit word of the row bitmap
        for (uint32_t jb = j0 + wave * 64; jb < j1; jb += SCREEN_THREADS) {
            const uint32_t j = jb + lane;
            bool pass = false;
            if (j < j1 && (!triangle || j > row)) {
                const uint32_t shared = cnt[j - j0];
                const uint32_t mr = rmeta[j].n_markers;
                const uint32_t mn = q.n_markers < mr ? q.n_markers : mr;
                pass = !screen_on || mn < ANI_SMALL_PASS || (double)shared > cutoff * (double)mn;
            }
            unsigned long long bits = __ballot(pass);
            if (lane == 0) {
                pass_bits[(uint64_t)blockIdx.x * words_per_row + (jb >> 6)] = bits;
                if (bits) atomicAdd(&s_rowcnt, (uint32_t)__popcll(bits));
            }
        }
        __syncthreads();
    }
    if (tid == 0) row_count[blockIdx.x] = s_rowcnt;
}

// expand the row bitmaps into an ordered pair list
__global__ __launch_bounds__(64) void pairs_fill_kernel(const unsigned long long *__restrict__ pass_bits, uint32_t words_per_row,
                                                        const uint32_t *__restrict__ rows, const uint32_t *__restrict__ row_off,
                                                        uint32_t *__restrict__ pair_q, uint32_t *__restrict__ pair_r)
{
    const uint32_t r = blockIdx.x, lane = threadIdx.x;
    uint32_t out = row_off[r];
    const uint32_t row = rows[r];
    for (uint32_t w0 = 0; w0 < words_per_row; w0 += 64) {
        uint32_t w = w0 + lane;
        unsigned long long bits = w < words_per_row ? pass_bits[(uint64_t)r * words_per_row + w] : 0ULL;
        uint32_t c = (uint32_t)__popcll(bits), total;
        uint32_t ex = wave_excl_scan(c, total);
        uint32_t o = out + ex;
        while (bits) {
            int b = __ffsll((long long)bits) - 1;
            bits &= bits - 1;
            pair_q[o] = row;
            pair_r[o] = w * 64 + b;
            o++;
        }
        out += total;
    }
}

// the slot of each query marker in the reference table (rectangle mode: the query set is not the
// set the table was built from)
__global__ __launch_bounds__(256) void table_lookup_kernel(const GenomeMeta *__restrict__ meta, const uint64_t *__restrict__ markers,
                                                           const unsigned long long *__restrict__ keys, uint64_t mask,
                                                           uint32_t *__restrict__ slot_of)
{
    const GenomeMeta m = meta[blockIdx.x];
    for (uint32_t e = threadIdx.x; e < m.n_markers; e += 256) {
        const uint64_t key = markers[m.marker_off + e];
        uint64_t slot = mix64(key) & mask;
        uint32_t res = 0xFFFFFFFFu;
        for (;;) {
            unsigned long long k = keys[slot];
            if (k == key) { res = (uint32_t)slot; break; }
            if (k == MK_EMPTY) break;
            slot = (slot + 1) & mask;
        }
        slot_of[m.marker_off + e] = res;
    }
}

// cutoff = (s/100)^marker_k by repeated multiplication, as the oracle's powi_fixed()
static double screen_cutoff(double screen_pct)
{
    double x = screen_pct / 100.0, r = 1.0;
    for (int i = 0; i < ANI_MARKER_K; i++) r = r * x;
    return r;
}

void screen_pairs(skder_sketches *refs, skder_sketches *queries, const std::vector<uint32_t> &rows, bool triangle,
                  double screen_pct, std::vector<uint32_t> &pair_q, std::vector<uint32_t> &pair_r)
{
    skder_ctx *ctx = refs->ctx;
    hipStream_t st = ctx->stream;
    pair_q.clear(); pair_r.clear();
    const uint32_t nrows = (uint32_t)rows.size(), nref = refs->n_genomes;
    if (!nrows || !nref) return;
    // pairs are counted and placed with 32-bit offsets: the callers (chain.hip rows_per_block) keep a call's worst case below 2^32
    if ((uint64_t)nrows * (uint64_t)nref >= (1ull << 32))
        throw SkError("screen: " + std::to_string(nrows) + " rows x " + std::to_string(nref) + " genomes exceed the 2^32 pairs of one screening call");
    const uint64_t total_marks = refs->h_marker_off[nref];
    ScreenIndex &X = refs->screen;
    ScanWorkspace ws;
    if (!X.built) {
        uint64_t ts = 1024;
        while (ts < 2 * total_marks) ts <<= 1;
        if (ts > 0x80000000ull) throw SkError("marker table too large");
        DevBuf<uint32_t> cnt, pos_of;
        X.keys.resize(ts, st); cnt.resize(ts + 1, st); X.loff.resize(ts + 1, st); pos_of.resize(total_marks + 1, st);
        X.slot_of.resize(total_marks + 1, st); X.list.resize(total_marks + 1, st);
        hipLaunchKernelGGL(table_clear_kernel, dim3(2048), dim3(256), 0, st, X.keys.p, cnt.p, ts);
        HIPCHECK(hipMemsetAsync(cnt.p + ts, 0, 4, st));
        hipLaunchKernelGGL(table_insert_kernel, dim3(nref), dim3(256), 0, st, refs->d_meta.p, refs->markers.p,
                           reinterpret_cast<unsigned long long *>(X.keys.p), cnt.p, ts - 1, X.slot_of.p, pos_of.p);
        exclusive_scan_u32(cnt.p, X.loff.p, ts + 1, ws, st);
        hipLaunchKernelGGL(table_fill_kernel, dim3(nref), dim3(256), 0, st, refs->d_meta.p, X.slot_of.p, X.loff.p, pos_of.p, X.list.p);
        X.ts = ts;
        X.built = true;
    }
    const uint64_t ts = X.ts;
    DevBuf<uint32_t> d_rows, row_count, row_off, q_slot;
    DevBuf<unsigned long long> pass_bits;
    DevBuf<uint64_t> &keys = X.keys;
    DevBuf<uint32_t> &loff = X.loff, &list = X.list, &slot_of = X.slot_of;
    const uint32_t *q_slot_ptr = slot_of.p;
    if (queries != refs) {
        q_slot.resize(queries->h_marker_off[queries->n_genomes] + 1, st);
        hipLaunchKernelGGL(table_lookup_kernel, dim3(queries->n_genomes), dim3(256), 0, st, queries->d_meta.p,
                           queries->markers.p, reinterpret_cast<const unsigned long long *>(keys.p), ts - 1, q_slot.p);
        q_slot_ptr = q_slot.p;
    }
    const uint32_t wpr = (nref + 63) / 64;
    d_rows.resize(nrows, st); row_count.resize(nrows + 1, st); row_off.resize(nrows + 1, st);
    pass_bits.resize((size_t)nrows * wpr, st);
    HIPCHECK(hipMemcpyAsync(d_rows.p, rows.data(), nrows * 4, hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemsetAsync(row_count.p + nrows, 0, 4, st));
    HIPCHECK(hipMemsetAsync(pass_bits.p, 0, (size_t)nrows * wpr * 8, st));
    const double cutoff = screen_cutoff(screen_pct);
    hipLaunchKernelGGL(screen_rows_kernel, dim3(nrows), dim3(SCREEN_THREADS), 0, st, queries->d_meta.p, q_slot_ptr, refs->d_meta.p, nref,
                       loff.p, list.p, d_rows.p, triangle ? 1 : 0, cutoff, screen_pct > 0.0 ? 1 : 0, pass_bits.p, wpr,
                       row_count.p);
    HIPCHECK(hipGetLastError());
    exclusive_scan_u32(row_count.p, row_off.p, nrows + 1, ws, st);
    uint32_t npairs = 0;
    HIPCHECK(hipMemcpyAsync(&npairs, row_off.p + nrows, 4, hipMemcpyDeviceToHost, st));
    HIPCHECK(hipStreamSynchronize(st));
    if (!npairs) return;
    DevBuf<uint32_t> d_pq, d_pr;
    d_pq.resize(npairs, st); d_pr.resize(npairs, st);
    hipLaunchKernelGGL(pairs_fill_kernel, dim3(nrows), dim3(64), 0, st, pass_bits.p, wpr, d_rows.p, row_off.p, d_pq.p, d_pr.p);
    pair_q.resize(npairs); pair_r.resize(npairs);
    HIPCHECK(hipMemcpyAsync(pair_q.data(), d_pq.p, npairs * 4ull, hipMemcpyDeviceToHost, st));
    HIPCHECK(hipMemcpyAsync(pair_r.data(), d_pr.p, npairs * 4ull, hipMemcpyDeviceToHost, st));
    HIPCHECK(hipStreamSynchronize(st));
}
