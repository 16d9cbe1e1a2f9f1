// index.hip -- per-genome lookup structures built from the position-ordered seeds:
//   * k-mer bucket index: seeds scattered into 2^bits multiplicative-hash buckets, every bucket
//     sorted by (k-mer, gpos) -- the "radix sort + dedup" stage of the sketch (one radix pass on a
//     hashed digit + in-bucket insertion sort; buckets hold ~4 seeds);
//   * repetitive k-mer cut-off (ani_oracle.c genome_finish);
//   * chunk id of every seed: (record, (gpos - record_off) / 20000) numbered in position order.
// One 1024-thread workgroup per genome (the 2^15 LDS counters allow one workgroup per CU, so the
// latency of the global-memory phases is hidden by 16 wavefronts); all counters live in LDS.
#include "device_utils.h"
#include "engine.h"

#define IDX_THREADS 1024
__global__ __launch_bounds__(IDX_THREADS) void index_genome_kernel(
    GenomeMeta *__restrict__ meta, const uint32_t *__restrict__ list, const uint32_t *__restrict__ rec_goff,
    const uint32_t *__restrict__ seed_kmer, const uint32_t *__restrict__ seed_gpos, const uint32_t *__restrict__ seed_ctg,
    uint32_t *__restrict__ skmer, uint32_t *__restrict__ sgpos, uint32_t *__restrict__ sctg, uint32_t *__restrict__ stag,
    uint32_t *__restrict__ boff_all, uint32_t *__restrict__ pchunk, uint32_t *__restrict__ chunk_start_all, uint8_t *__restrict__ pcs)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint32_t *cnt = reinterpret_cast<uint32_t *>(smem_raw);   // 2^bits counters, later cursor, later histogram
    __shared__ uint32_t wsum[IDX_THREADS / 64];
    __shared__ uint32_t s_distinct;

    const uint32_t g = list[blockIdx.x], tid = threadIdx.x;
    GenomeMeta m = meta[g];
    const uint32_t n = m.n_seeds, bits = m.bucket_bits, nb = 1u << bits;
    const uint32_t *pk = seed_kmer + m.seed_off, *pg = seed_gpos + m.seed_off, *pc = seed_ctg + m.seed_off;
    uint32_t *ok = skmer + m.seed_off, *og = sgpos + m.seed_off, *oc = sctg + m.seed_off, *ot = stag + m.seed_off;
    uint32_t *boff = boff_all + m.bucket_off;
    const uint32_t *rg = rec_goff + m.rec_goff_off;

    for (uint32_t b = tid; b < nb; b += IDX_THREADS) cnt[b] = 0;
    if (tid == 0) s_distinct = 0;
    __syncthreads();
    for (uint32_t s = tid; s < n; s += IDX_THREADS) atomicAdd(&cnt[kmer_bucket(pk[s] & SK_SEED_MASK, bits)], 1u);
    __syncthreads();
    // exclusive scan of the bucket counts -> boff (global); counters are reset for the scatter cursor
    uint32_t running = 0;
    for (uint32_t base = 0; base < nb; base += IDX_THREADS) {
        uint32_t b = base + tid;
        uint32_t v = b < nb ? cnt[b] : 0u;
        uint32_t total;
        uint32_t ex = block_excl_scan<IDX_THREADS / 64>(v, wsum, total);
        if (b < nb) { boff[b] = running + ex; cnt[b] = 0; }
        running += total;
    }
    if (tid == 0) boff[nb] = n;
    __syncthreads();
    for (uint32_t s = tid; s < n; s += IDX_THREADS) {
        uint32_t km = pk[s];
        uint32_t b = kmer_bucket(km & SK_SEED_MASK, bits);
        uint32_t pos = boff[b] + atomicAdd(&cnt[b], 1u);
        ok[pos] = km; og[pos] = pg[s]; oc[pos] = pc[s];
    }
    __syncthreads();   // global writes of this workgroup are visible to it after the barrier
    // histogram of multiplicities reuses the counter array
    for (uint32_t b = tid; b < IDX_REP_HIST; b += IDX_THREADS) cnt[b] = 0;
    __syncthreads();
    uint32_t my_distinct = 0;
    for (uint32_t b = tid; b < nb; b += IDX_THREADS) {
        const uint32_t lo = boff[b], hi = (b + 1 == nb) ? n : boff[b + 1];
        // insertion sort by (kmer, gpos); gpos is unique inside a genome, so the order is total
        for (uint32_t i = lo + 1; i < hi; i++) {
            uint32_t km = ok[i], gp = og[i], ct = oc[i];
            uint32_t kk = km & SK_SEED_MASK;
            uint32_t j = i;
            while (j > lo) {
                uint32_t pk2 = ok[j - 1] & SK_SEED_MASK;
                if (pk2 < kk || (pk2 == kk && og[j - 1] < gp)) break;
                ok[j] = ok[j - 1]; og[j] = og[j - 1]; oc[j] = oc[j - 1];
                j--;
            }
            ok[j] = km; og[j] = gp; oc[j] = ct;
        }
        for (uint32_t i = lo; i < hi;) {
            uint32_t kk = ok[i] & SK_SEED_MASK, j = i + 1;
            while (j < hi && (ok[j] & SK_SEED_MASK) == kk) j++;
            uint32_t mult = j - i;
            atomicAdd(&cnt[mult < IDX_REP_HIST - 1 ? mult : IDX_REP_HIST - 1], 1u);
            my_distinct++;
            i = j;
        }
    }
    atomicAdd(&s_distinct, my_distinct);
    __syncthreads();
    for (uint32_t e = tid; e < n; e += IDX_THREADS) ot[e] = og[e] | ((oc[e] & 63u) << HIT_POS_BITS) | (ok[e] & 0x80000000u);
    if (tid == 0) {
        // multiplicity of ascending rank D - D/1000 - 1 == the (D/1000 + 1)-th largest
        uint32_t D = s_distinct, cut = 0xFFFFFFFFu;
        if (D) {
            uint32_t need = D / 1000u + 1u, cum = 0, mval = 0;
            for (int mm = IDX_REP_HIST - 1; mm >= 1; mm--) {
                cum += cnt[mm];
                if (cum >= need) { mval = (uint32_t)mm; break; }
            }
            if (mval >= ANI_REP_FLOOR) cut = mval;
        }
        meta[g].rep_cut = cut;
    }
    // chunk ids in position order
    uint32_t crun = 0;
    for (uint32_t base = 0; base < n; base += IDX_THREADS) {
        uint32_t s = base + tid;
        uint32_t flag = 0;
        if (s < n) {
            uint32_t c = pc[s], ck = (pg[s] - rg[c]) / ANI_CHUNK_LEN;
            if (s == 0) flag = 1;
            else {
                uint32_t c2 = pc[s - 1], ck2 = (pg[s - 1] - rg[c2]) / ANI_CHUNK_LEN;
                flag = (c != c2) || (ck != ck2);
            }
        }
        uint32_t total;
        uint32_t ex = block_excl_scan<IDX_THREADS / 64>(flag, wsum, total);
        if (s < n) {
            pchunk[m.seed_off + s] = crun + ex + flag - 1u;
            pcs[m.seed_off + s] = (uint8_t)flag;
            if (flag) chunk_start_all[m.chunk_off + crun + ex] = s;
        }
        crun += total;
    }
    if (tid == 0) { meta[g].n_chunks = crun; chunk_start_all[m.chunk_off + crun] = n; }
}

// ---------------------------------------------------------------------------------------------
// LDS-resident variant for genomes with fewer than 65536 seeds whose tables fit (the normal case):
// 16-bit bucket counters/cursors packed two per word, a 16-bit permutation of the seed indices, and
// the multiplicity histogram all live in LDS; global memory sees only coalesced reads of the
// position-ordered arrays, gathers through the finished permutation and coalesced writes.
#define IDXF_THREADS 1024
#ifndef IDX_PACKED
#define IDX_PACKED 1            // 1: a (k-mer, position, record) copy of the seeds written in phase A and gathered in D / E with ONE request per seed
#endif
#define IDXF_U 8                // independent loads per thread and trip
#define IDXF_FIXED_BYTES (IDX_REP_HIST * 4)
__host__ __device__ inline size_t idxf_smem_bytes(uint32_t nb, uint32_t n)
{
    size_t body = (size_t)nb * 2 + (((size_t)n * 2 + 15) & ~(size_t)15);   // counters/cursors + permutation
    if (body < 12 * IDXF_THREADS) body = 12 * IDXF_THREADS;                   // phase F: 1024 ballots + 1024 offsets
    return body + IDXF_FIXED_BYTES;
}
static_assert(IDX_REP_HIST == 4 * IDXF_THREADS, "rep-cut scan assumes four histogram bins per thread");

__global__ __launch_bounds__(IDXF_THREADS) void index_genome_lds_kernel(
    GenomeMeta *__restrict__ meta, const uint32_t *__restrict__ list, const uint32_t *__restrict__ rec_goff,
    const uint32_t *__restrict__ seed_kmer, const uint32_t *__restrict__ seed_gpos, const uint32_t *__restrict__ seed_ctg,
    uint32_t *__restrict__ skmer, uint32_t *__restrict__ sgpos, uint32_t *__restrict__ sctg, uint32_t *__restrict__ stag,
    uint32_t *__restrict__ boff_all, uint32_t *__restrict__ pchunk, uint32_t *__restrict__ chunk_start_all,
    uint4 *__restrict__ packed_all, uint8_t *__restrict__ pcs)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ uint32_t wsum[IDXF_THREADS / 64];
    __shared__ uint32_t s_distinct, s_cut;
    const uint32_t g = list[blockIdx.x], tid = threadIdx.x;
    const GenomeMeta m = meta[g];
    const uint32_t n = m.n_seeds, bits = m.bucket_bits, nb = 1u << bits;
    uint32_t *hist = reinterpret_cast<uint32_t *>(smem_raw);                       // IDX_REP_HIST
    uint32_t *cntp = hist + IDX_REP_HIST;                                          // nb / 2 words: two 16-bit fields each
    uint16_t *cur16 = reinterpret_cast<uint16_t *>(cntp);                          // the same fields, one per bucket
    uint16_t *perm = reinterpret_cast<uint16_t *>(cntp + nb / 2);                  // n
    const uint32_t *pk = seed_kmer + m.seed_off, *pg = seed_gpos + m.seed_off, *pc = seed_ctg + m.seed_off;
    uint32_t *ok = skmer + m.seed_off, *og = sgpos + m.seed_off, *oc = sctg + m.seed_off, *ot = stag + m.seed_off;
    uint32_t *boff = boff_all + m.bucket_off;
    const uint32_t *rg = rec_goff + m.rec_goff_off;
    uint4 *packed = packed_all + m.seed_off;      // written in A, gathered in E: one 16-byte request per seed instead of three
    (void)packed;

    for (uint32_t b = tid; b < nb / 2; b += IDXF_THREADS) cntp[b] = 0;
    for (uint32_t b = tid; b < IDX_REP_HIST; b += IDXF_THREADS) hist[b] = 0;
    if (tid == 0) s_distinct = 0;
    __syncthreads();
    // A. bucket sizes.  (Here and below: IDXF_U loads in flight per thread before anything depends on them --
    // the kernel is a chain of short phases whose cost is the memory latency they expose.)
    for (uint32_t s0 = tid; s0 < n; s0 += IDXF_U * IDXF_THREADS) {
#if IDX_PACKED
        uint32_t kv[IDXF_U], gv[IDXF_U], cv[IDXF_U];
#pragma unroll
        for (int u = 0; u < IDXF_U; u++) {
            const uint32_t s = s0 + u * IDXF_THREADS;
            kv[u] = s < n ? pk[s] : 0u; gv[u] = s < n ? pg[s] : 0u; cv[u] = s < n ? pc[s] : 0u;
        }
#else
        uint32_t kv[IDXF_U];
#pragma unroll
        for (int u = 0; u < IDXF_U; u++) { const uint32_t s = s0 + u * IDXF_THREADS; kv[u] = s < n ? pk[s] : 0u; }
#endif
#pragma unroll
        for (int u = 0; u < IDXF_U; u++) {
            const uint32_t s = s0 + u * IDXF_THREADS;
            if (s < n) {
#if IDX_PACKED
                packed[s] = make_uint4(kv[u], gv[u], cv[u], 0u);
#endif
                const uint32_t b = kmer_bucket(kv[u] & SK_SEED_MASK, bits);
                atomicAdd(&cntp[b >> 1], 1u << ((b & 1u) * 16u));
            }
        }
    }
    __syncthreads();
    // B. exclusive scan: every thread owns nb/1024 consecutive 16-bit fields, handled as whole words
    // (16-byte LDS accesses); sizes become cursors.  Tables with fewer than 2048 buckets: one word per
    // thread.
    {
        const uint32_t nw = nb / 2;                                   // words
        const uint32_t wper = (nw + IDXF_THREADS - 1) / IDXF_THREADS; // 16 for 2^15 buckets
        const uint32_t w0 = tid * wper < nw ? tid * wper : nw, w1 = w0 + wper < nw ? w0 + wper : nw;
        uint32_t loc = 0;
        if (wper % 4 == 0) {
            for (uint32_t w = w0; w < w1; w += 4) {
                const uint4 v = *reinterpret_cast<const uint4 *>(cntp + w);
                loc += (v.x & 0xFFFFu) + (v.x >> 16) + (v.y & 0xFFFFu) + (v.y >> 16) + (v.z & 0xFFFFu) + (v.z >> 16) + (v.w & 0xFFFFu) + (v.w >> 16);
            }
        } else {
            for (uint32_t w = w0; w < w1; w++) { const uint32_t v = cntp[w]; loc += (v & 0xFFFFu) + (v >> 16); }
        }
        uint32_t total;
        uint32_t run = block_excl_scan<IDXF_THREADS / 64>(loc, wsum, total);
        auto step = [&run](uint32_t v) {          // two sizes -> two cursors
            const uint32_t c0 = v & 0xFFFFu, c1 = v >> 16;
            const uint32_t r = run | ((run + c0) << 16);
            run += c0 + c1;
            return r;
        };
        if (wper % 4 == 0) {
            for (uint32_t w = w0; w < w1; w += 4) {
                uint4 v = *reinterpret_cast<const uint4 *>(cntp + w);
                v.x = step(v.x); v.y = step(v.y); v.z = step(v.z); v.w = step(v.w);
                *reinterpret_cast<uint4 *>(cntp + w) = v;
            }
        } else {
            for (uint32_t w = w0; w < w1; w++) cntp[w] = step(cntp[w]);
        }
    }
    __syncthreads();
    for (uint32_t b = tid; b < nb; b += IDXF_THREADS) boff[b] = cur16[b];
    if (tid == 0) boff[nb] = n;
    __syncthreads();
    // C. permutation: seed indices grouped by bucket (order inside a bucket fixed in D).  A cursor ends
    // at the start of the next bucket (<= n < 65536), so a field never carries into its neighbour
    for (uint32_t s0 = tid; s0 < n; s0 += IDXF_U * IDXF_THREADS) {
        uint32_t kv[IDXF_U];
#pragma unroll
        for (int u = 0; u < IDXF_U; u++) { const uint32_t s = s0 + u * IDXF_THREADS; kv[u] = s < n ? pk[s] : 0u; }
#pragma unroll
        for (int u = 0; u < IDXF_U; u++) {
            const uint32_t s = s0 + u * IDXF_THREADS;
            if (s < n) {
                const uint32_t b = kmer_bucket(kv[u] & SK_SEED_MASK, bits), sh = (b & 1u) * 16u;
                const uint32_t pos = (atomicAdd(&cntp[b >> 1], 1u << sh) >> sh) & 0xFFFFu;
                perm[pos] = (uint16_t)s;
            }
        }
    }
    __syncthreads();
    // D. order inside every bucket by (k-mer, gpos); multiplicity histogram.  Buckets are independent:
    // strided over the threads (cursor b now holds the END of bucket b)
    uint32_t my_distinct = 0, ones = 0;
    for (uint32_t b = tid; b < nb; b += IDXF_THREADS) {
        const uint32_t lo = b ? cur16[b - 1] : 0u, hi = cur16[b], k = hi - lo;
        if (k == 1) { ones++; }
        else if (k >= 2 && k <= 8) {
            // buckets of 2..8 seeds: all k-mers fetched at once (a wavefront always has some lane here, and
            // an insertion sort through dependent gathers made every lane wait for the longest bucket),
            // sorted in registers by (k-mer, seed index) -- seeds are in position order, so the seed index
            // orders equal k-mers by position
            uint64_t key[9];
            // the longest bucket among the lanes here decides how many entries are fetched at all (mostly 3 or 4 of the 8)
            uint32_t kmax = 2;
#pragma unroll
            for (int i = 2; i < 8; i++) if (__any(k > (uint32_t)i)) kmax = (uint32_t)i + 1u;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                key[i] = ~0ull;
                if ((uint32_t)i < kmax) {       // wave-uniform
                    const bool in = (uint32_t)i < k;
                    const uint32_t si = in ? perm[lo + i] : 0u;
#if IDX_PACKED
                    const uint32_t kk = packed[si].x & SK_SEED_MASK;
#else
                    const uint32_t kk = pk[si] & SK_SEED_MASK;
#endif
                    key[i] = in ? (((uint64_t)kk << 16) | si) : ~0ull;
                }
            }
            key[8] = ~0ull;
#define CE(A_, B_) { const uint64_t x_ = key[A_], y_ = key[B_]; const bool sw_ = y_ < x_; key[A_] = sw_ ? y_ : x_; key[B_] = sw_ ? x_ : y_; }
            CE(0, 1) CE(2, 3) CE(4, 5) CE(6, 7)
            CE(0, 2) CE(1, 3) CE(4, 6) CE(5, 7)
            CE(1, 2) CE(5, 6)
            CE(0, 4) CE(1, 5) CE(2, 6) CE(3, 7)
            CE(2, 4) CE(3, 5)
            CE(1, 2) CE(3, 4) CE(5, 6)
#undef CE
            uint32_t mult = 1;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if ((uint32_t)i < k) perm[lo + i] = (uint16_t)(key[i] & 0xFFFFu);
                // run of equal k-mers ending at entry i?
                const bool last = (uint32_t)i + 1u == k;
                const bool same = (uint32_t)i + 1u < k && (uint32_t)(key[i + 1] >> 16) == (uint32_t)(key[i] >> 16);
                if ((uint32_t)i < k) {
                    if (same) mult++;
                    else {
                        if (mult == 1) ones++;
                        else { atomicAdd(&hist[mult], 1u); my_distinct++; }
                        mult = 1;
                    }
                }
                (void)last;
            }
        } else if (k > 8) {
            for (uint32_t i = lo + 1; i < hi; i++) {
                const uint32_t si = perm[i], kk = pk[si] & SK_SEED_MASK, gp = pg[si];
                uint32_t j = i;
                while (j > lo) {
                    const uint32_t sj = perm[j - 1], k2 = pk[sj] & SK_SEED_MASK;
                    if (k2 < kk || (k2 == kk && pg[sj] < gp)) break;
                    perm[j] = (uint16_t)sj;
                    j--;
                }
                perm[j] = (uint16_t)si;
            }
            for (uint32_t i = lo; i < hi;) {
                const uint32_t kk = pk[perm[i]] & SK_SEED_MASK;
                uint32_t j = i + 1;
                while (j < hi && (pk[perm[j]] & SK_SEED_MASK) == kk) j++;
                const uint32_t mult = j - i;
                if (mult == 1) ones++;
                else { atomicAdd(&hist[mult < IDX_REP_HIST - 1 ? mult : IDX_REP_HIST - 1], 1u); my_distinct++; }
                i = j;
            }
        }
    }
    if (ones) atomicAdd(&hist[1], ones);
    atomicAdd(&s_distinct, my_distinct + ones);
    __syncthreads();
    // repetitive cut-off: multiplicity of ascending rank D - D/1000 - 1 == the (D/1000 + 1)-th largest.
    // Thread t sums bins [4 (T-1-t), +4): a scan in thread order runs from the highest multiplicity down
    {
        const uint32_t D = s_distinct, need = D / 1000u + 1u;
        const uint32_t hb = 4u * (IDXF_THREADS - 1u - tid);
        const uint32_t c3 = hist[hb + 3], c2 = hist[hb + 2], c1 = hist[hb + 1], c0 = hb ? hist[hb] : 0u;
        uint32_t total;
        const uint32_t before = block_excl_scan<IDXF_THREADS / 64>(c3 + c2 + c1 + c0, wsum, total);
        if (tid == 0) s_cut = 0;
        __syncthreads();
        if (D && before < need && before + c3 + c2 + c1 + c0 >= need) {
            uint32_t cum = before, mval = 0;
            cum += c3; if (!mval && cum >= need) mval = hb + 3;
            cum += c2; if (!mval && cum >= need) mval = hb + 2;
            cum += c1; if (!mval && cum >= need) mval = hb + 1;
            cum += c0; if (!mval && cum >= need) mval = hb;
            s_cut = mval;
        }
        __syncthreads();
        if (tid == 0) meta[g].rep_cut = (D && s_cut >= ANI_REP_FLOOR) ? s_cut : 0xFFFFFFFFu;
    }
    // E. bucket-ordered arrays: gathers through the permutation, coalesced writes
    for (uint32_t p0 = tid; p0 < n; p0 += IDXF_U * IDXF_THREADS) {
        uint4 pv[IDXF_U];
#pragma unroll
        for (int u = 0; u < IDXF_U; u++) {
            const uint32_t pos = p0 + u * IDXF_THREADS;
#if IDX_PACKED
            pv[u] = packed[pos < n ? perm[pos] : 0u];
#else
            const uint32_t si = pos < n ? perm[pos] : 0u;
            pv[u] = make_uint4(pk[si], pg[si], pc[si], 0u);
#endif
        }
#pragma unroll
        for (int u = 0; u < IDXF_U; u++) {
            const uint32_t pos = p0 + u * IDXF_THREADS;
            if (pos < n) {
                ok[pos] = pv[u].x; og[pos] = pv[u].y; oc[pos] = pv[u].z;
                ot[pos] = pv[u].y | ((pv[u].z & 63u) << HIT_POS_BITS) | (pv[u].x & 0x80000000u);
            }
        }
    }
    __syncthreads();   // the permutation is dead from here on
    // F. chunk ids in position order.  Seed s starts a chunk when its (record, 20 kb window) differs
    // from seed s-1's.  Coalesced passes: ballots of the start flags per 64 seeds (kept in the counter
    // area, free by now), one scan over the per-ballot counts, ids from ballot prefixes.
    {
        unsigned long long *ballots = reinterpret_cast<unsigned long long *>(cntp);     // <= 1024 (n < 65536)
        uint32_t *blk_off = reinterpret_cast<uint32_t *>(ballots + IDXF_THREADS);      // 1024
        const uint32_t nblk = (n + 63u) / 64u, lane = tid & 63u;
        for (uint32_t base = 0; base < nblk * 64u; base += IDXF_U * IDXF_THREADS) {
            // the loads of IDXF_U trips first (record, position, the record's start), then the ballots
            uint32_t c1[IDXF_U], c2[IDXF_U], g1[IDXF_U], g2[IDXF_U], r1[IDXF_U], r2[IDXF_U];
#pragma unroll
            for (int u = 0; u < IDXF_U; u++) {
                const uint32_t sx = base + u * IDXF_THREADS + tid;
                const bool in = sx < n && sx > 0;
                c1[u] = in ? pc[sx] : 0u; c2[u] = in ? pc[sx - 1] : 0u;
                g1[u] = in ? pg[sx] : 0u; g2[u] = in ? pg[sx - 1] : 0u;
            }
#pragma unroll
            for (int u = 0; u < IDXF_U; u++) { r1[u] = rg[c1[u]]; r2[u] = rg[c2[u]]; }
#pragma unroll
            for (int u = 0; u < IDXF_U; u++) {
                const uint32_t sx = base + u * IDXF_THREADS + tid;
                const bool flag = sx < n && (sx == 0 || c1[u] != c2[u] || (g1[u] - r1[u]) / ANI_CHUNK_LEN != (g2[u] - r2[u]) / ANI_CHUNK_LEN);
                const unsigned long long bal = __ballot(flag);
                if (lane == 0 && sx / 64u < nblk) ballots[sx / 64u] = bal;
            }
        }
        __syncthreads();
        uint32_t total;
        const uint32_t mine = tid < nblk ? (uint32_t)__popcll(ballots[tid]) : 0u;
        const uint32_t ex = block_excl_scan<IDXF_THREADS / 64>(mine, wsum, total);
        if (tid < nblk) blk_off[tid] = ex;
        __syncthreads();
        for (uint32_t s = tid; s < n; s += IDXF_THREADS) {
            const unsigned long long bal = ballots[s / 64u];
            const uint32_t upto = (uint32_t)__popcll(bal & (~0ull >> (63u - lane)));   // flags at lanes <= mine
            const uint32_t id = blk_off[s / 64u] + upto;                                // chunks started up to and including s
            pchunk[m.seed_off + s] = id - 1u;
            pcs[m.seed_off + s] = (uint8_t)((bal >> lane) & 1ull);
            if ((bal >> lane) & 1ull) chunk_start_all[m.chunk_off + id - 1u] = s;
        }
        if (tid == 0) { meta[g].n_chunks = total; chunk_start_all[m.chunk_off + total] = n; }
    }
}

// chunk tables only (chunk id and chunk-start flag of every seed, first seed of every chunk): what a genome needs as the
// CHUNKED side of a pair.  Multi-GPU runs build the bucket index of a genome only on the GPU that owns it (the pairs that
// probe it are chained there) and these tables everywhere.  rep_cut is left alone: the owner's value is installed by the
// caller (index_set_rep_cuts).
__global__ __launch_bounds__(IDX_THREADS) void chunk_tables_kernel(
    GenomeMeta *__restrict__ meta, const uint32_t *__restrict__ list, const uint32_t *__restrict__ rec_goff,
    const uint32_t *__restrict__ seed_gpos, const uint32_t *__restrict__ seed_ctg,
    uint32_t *__restrict__ pchunk, uint32_t *__restrict__ chunk_start_all, uint8_t *__restrict__ pcs)
{
    __shared__ uint32_t wsum[IDX_THREADS / 64];
    const uint32_t g = list[blockIdx.x], tid = threadIdx.x;
    const GenomeMeta m = meta[g];
    const uint32_t n = m.n_seeds;
    const uint32_t *pg = seed_gpos + m.seed_off, *pc = seed_ctg + m.seed_off;
    const uint32_t *rg = rec_goff + m.rec_goff_off;
    uint32_t crun = 0;
    for (uint32_t base = 0; base < n; base += IDX_THREADS) {
        const uint32_t s = base + tid;
        uint32_t flag = 0;
        if (s < n) {
            const uint32_t c = pc[s], ck = (pg[s] - rg[c]) / ANI_CHUNK_LEN;
            if (s == 0) flag = 1;
            else {
                const uint32_t c2 = pc[s - 1], ck2 = (pg[s - 1] - rg[c2]) / ANI_CHUNK_LEN;
                flag = (c != c2) || (ck != ck2);
            }
        }
        uint32_t total;
        const uint32_t ex = block_excl_scan<IDX_THREADS / 64>(flag, wsum, total);
        if (s < n) {
            pchunk[m.seed_off + s] = crun + ex + flag - 1u;
            pcs[m.seed_off + s] = (uint8_t)flag;
            if (flag) chunk_start_all[m.chunk_off + crun + ex] = s;
        }
        crun += total;
    }
    if (tid == 0) { meta[g].n_chunks = crun; chunk_start_all[m.chunk_off + crun] = n; }
}

void index_impl(skder_sketches *s)
{
    if (s->indexed) return;
    if (!s->index_pending) index_begin(s, s->ctx->stream);
    index_finish(s);
}

// index kernels for the genomes of `which`: the full index where full_index says so, chunk tables otherwise
static void index_launch(skder_sketches *s, const std::vector<uint32_t> &which, hipStream_t st)
{
    // genomes whose tables fit in LDS take the LDS-resident kernel, the others the general one
    const size_t lds_limit = 150 * 1024;
    std::vector<uint32_t> &small = s->idx_small, &big = s->idx_big;
    std::vector<uint32_t> light;
    small.clear(); big.clear();
    size_t small_bytes = 0;
    uint32_t max_bits = 12;   // the general kernel's multiplicity histogram needs 4096 counters
    for (uint32_t g : which) {
        const GenomeMeta &m = s->h_meta[g];
        if (!s->full_index[g]) { light.push_back(g); continue; }
        const size_t need = idxf_smem_bytes(1u << m.bucket_bits, m.n_seeds);
        if (m.n_seeds < 65536u && need <= lds_limit) {
            small.push_back(g);
            small_bytes = need > small_bytes ? need : small_bytes;
        } else {
            big.push_back(g);
            max_bits = m.bucket_bits > max_bits ? m.bucket_bits : max_bits;
        }
    }
    DevBuf<uint32_t> &d_list = s->idx_list;
    d_list.resize(which.size() + 1, st);
    if (!small.empty()) HIPCHECK(hipMemcpyAsync(d_list.p, small.data(), small.size() * 4, hipMemcpyHostToDevice, st));
    if (!big.empty()) HIPCHECK(hipMemcpyAsync(d_list.p + small.size(), big.data(), big.size() * 4, hipMemcpyHostToDevice, st));
    if (!light.empty())
        HIPCHECK(hipMemcpyAsync(d_list.p + small.size() + big.size(), light.data(), light.size() * 4, hipMemcpyHostToDevice, st));
    if (!small.empty()) {
        HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(index_genome_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds_limit));
        hipLaunchKernelGGL(index_genome_lds_kernel, dim3((unsigned)small.size()), dim3(IDXF_THREADS), small_bytes, st, s->d_meta.p, d_list.p,
                           s->d_rec_goff.p, s->seed_kmer.p, s->seed_gpos.p, s->seed_ctg.p, s->skmer.p, s->sgpos.p, s->sctg.p, s->stag.p, s->boff.p,
                           s->pchunk.p, s->chunk_start.p, s->idx_packed.p, s->pcs.p);
    }
    if (!big.empty()) {
        HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(index_genome_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)((1u << max_bits) * 4)));
        hipLaunchKernelGGL(index_genome_kernel, dim3((unsigned)big.size()), dim3(IDX_THREADS), (1u << max_bits) * 4, st, s->d_meta.p,
                           d_list.p + small.size(), s->d_rec_goff.p, s->seed_kmer.p, s->seed_gpos.p, s->seed_ctg.p, s->skmer.p, s->sgpos.p,
                           s->sctg.p, s->stag.p, s->boff.p, s->pchunk.p, s->chunk_start.p, s->pcs.p);
    }
    if (!light.empty())
        hipLaunchKernelGGL(chunk_tables_kernel, dim3((unsigned)light.size()), dim3(IDX_THREADS), 0, st, s->d_meta.p,
                           d_list.p + small.size() + big.size(), s->d_rec_goff.p, s->seed_gpos.p, s->seed_ctg.p, s->pchunk.p, s->chunk_start.p,
                           s->pcs.p);
}

void index_begin(skder_sketches *s, hipStream_t st, const uint8_t *full)
{
    if (s->indexed || s->index_pending) return;
    skder_ctx *ctx = s->ctx;
    s->idx_stream = st;
    const uint32_t G = s->n_genomes;
    s->h_meta.resize(G);
    uint64_t boff_total = 0, rg = 0, chunk_total = 0;
    for (uint32_t g = 0; g < G; g++) {
        GenomeMeta &m = s->h_meta[g];
        m.seed_off = s->h_seed_off[g];
        m.n_seeds = (uint32_t)(s->h_seed_off[g + 1] - s->h_seed_off[g]);
        m.marker_off = s->h_marker_off[g];
        m.n_markers = (uint32_t)(s->h_marker_off[g + 1] - s->h_marker_off[g]);
        m.total_len = s->h_genome_len[g];
        m.n_rec = s->h_genome_nrec[g];
        m.rec_goff_off = rg;
        rg += m.n_rec + 1;
        uint32_t bits = 4;
        while (bits < IDX_MAX_BUCKET_BITS && (1u << bits) * 2u < m.n_seeds) bits++;   // 1-2 seeds per bucket
        m.bucket_bits = bits;
        m.bucket_off = boff_total;
        boff_total += (1u << bits) + 1;
        m.n_chunks = 0;
        m.rep_cut = 0xFFFFFFFFu;
        m.chunk_off = chunk_total;
        chunk_total += m.total_len / ANI_CHUNK_LEN + m.n_rec + 2;   // upper bound on chunks + sentinel
    }
    const uint64_t ns = s->h_seed_off[G];
    s->d_meta.resize(G, st);
    s->d_rec_goff.resize(s->h_rec_goff.size() + 1, st);
    s->skmer.resize(ns + 1, st); s->sgpos.resize(ns + 1, st); s->sctg.resize(ns + 1, st); s->stag.resize(ns + 1, st);
    s->pchunk.resize(ns + 1, st);
    s->pcs.resize(ns + 16, st);
    s->boff.resize(boff_total + 1, st);
    s->chunk_start.resize(chunk_total + 1, st);
    if (G) {
        // the genome table goes through ctx->stream: the marker screen, which runs there beside the index build,
        // reads it as well, and the sketches were produced there
        HIPCHECK(hipMemcpyAsync(s->d_meta.p, s->h_meta.data(), G * sizeof(GenomeMeta), hipMemcpyHostToDevice, ctx->stream));
        HIPCHECK(hipMemcpyAsync(s->d_rec_goff.p, s->h_rec_goff.data(), s->h_rec_goff.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        if (st != ctx->stream) {
            HIPCHECK(hipEventRecord(ctx->ev[11], ctx->stream));
            HIPCHECK(hipStreamWaitEvent(st, ctx->ev[11], 0));
        }
        HIPCHECK(hipEventRecord(ctx->ev[3], st));
        s->full_index.assign(G, 1);
        s->partial_index = 0;
        if (full) for (uint32_t g = 0; g < G; g++) { s->full_index[g] = full[g] ? 1 : 0; s->partial_index += full[g] ? 0u : 1u; }
        std::vector<uint32_t> all(G);
        for (uint32_t g = 0; g < G; g++) all[g] = g;
        s->idx_packed.resize(ns + 1, st);
        index_launch(s, all, st);
        HIPCHECK(hipGetLastError());
        HIPCHECK(hipEventRecord(ctx->ev[4], st));
    }
    s->index_pending = true;
}

void index_finish(skder_sketches *s)
{
    if (s->indexed || !s->index_pending) return;
    skder_ctx *ctx = s->ctx;
    if (s->n_genomes) {
        // (a copy into pageable memory would hold the host until the kernels are done: it is issued here, not in index_begin)
        HIPCHECK(hipMemcpyAsync(s->h_meta.data(), s->d_meta.p, s->n_genomes * sizeof(GenomeMeta), hipMemcpyDeviceToHost, s->idx_stream));
        HIPCHECK(hipStreamSynchronize(s->idx_stream));
        float ms = 0;
        HIPCHECK(hipEventElapsedTime(&ms, ctx->ev[3], ctx->ev[4]));
        ctx->timing_index = ms;
    }
    s->idx_list.release();
    s->idx_packed.release();
    s->index_pending = false;
    s->indexed = true;
}

// full index for genomes that so far have chunk tables only (a chunked genome whose own repetitive-k-mer filter is active
// needs its bucket index on the slow chaining path); complete on return
void index_promote(skder_sketches *s, const std::vector<uint32_t> &genomes)
{
    if (!s->indexed) throw SkError("index_promote: the set is not indexed");
    std::vector<uint32_t> todo;
    for (uint32_t g : genomes)
        if (g < s->n_genomes && !s->full_index[g]) { s->full_index[g] = 1; s->partial_index--; todo.push_back(g); }
    if (todo.empty()) return;
    hipStream_t st = s->ctx->stream;
    const uint64_t ns = s->h_seed_off[s->n_genomes];
    // the kernels overwrite rep_cut with what they compute (the same value the owner found) and n_chunks (unchanged)
    s->idx_packed.resize(ns + 1, st);
    index_launch(s, todo, st);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipMemcpyAsync(s->h_meta.data(), s->d_meta.p, s->n_genomes * sizeof(GenomeMeta), hipMemcpyDeviceToHost, st));
    HIPCHECK(hipStreamSynchronize(st));
    s->idx_list.release();
    s->idx_packed.release();
}

// repetitive-k-mer cut-offs found by the owners of the genomes this set holds chunk tables of (mask[g] != 0: install in[g])
void index_set_rep_cuts(skder_sketches *s, const uint32_t *in, const uint8_t *mask)
{
    if (!s->indexed) throw SkError("index_set_rep_cuts: the set is not indexed");
    for (uint32_t g = 0; g < s->n_genomes; g++)
        if (!mask || mask[g]) s->h_meta[g].rep_cut = in[g];
    HIPCHECK(hipMemcpyAsync(s->d_meta.p, s->h_meta.data(), s->n_genomes * sizeof(GenomeMeta), hipMemcpyHostToDevice, s->ctx->stream));
    HIPCHECK(hipStreamSynchronize(s->ctx->stream));
}
