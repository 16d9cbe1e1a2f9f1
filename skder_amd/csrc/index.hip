// index.hip -- per-genome lookup structures built from the position-ordered seeds:
//   * k-mer bucket index: seeds scattered into 2^bits multiplicative-hash buckets, every bucket
//     sorted by (k-mer, gpos) -- the "radix sort + dedup" stage of the sketch (one radix pass on a
//     hashed digit + in-bucket insertion sort; buckets hold ~4 seeds);
//   * repetitive k-mer cut-off (ani_oracle.c genome_finish);
//   * chunk id of every seed: (record, (gpos - record_off) / 20000) numbered in position order.
// One 256-thread workgroup per genome; all counters live in LDS.
#include "device_utils.h"
#include "engine.h"

__global__ __launch_bounds__(256) void index_genome_kernel(
    GenomeMeta *__restrict__ meta, const uint32_t *__restrict__ rec_goff, const uint32_t *__restrict__ seed_kmer,
    const uint32_t *__restrict__ seed_gpos, const uint32_t *__restrict__ seed_ctg, uint32_t *__restrict__ skmer,
    uint32_t *__restrict__ sgpos, uint32_t *__restrict__ sctg, uint32_t *__restrict__ boff_all,
    uint32_t *__restrict__ pchunk, uint32_t *__restrict__ chunk_start_all)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint32_t *cnt = reinterpret_cast<uint32_t *>(smem_raw);   // 2^bits counters, later cursor, later histogram
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t s_distinct;

    const uint32_t g = blockIdx.x, tid = threadIdx.x;
    GenomeMeta m = meta[g];
    const uint32_t n = m.n_seeds, bits = m.bucket_bits, nb = 1u << bits;
    const uint32_t *pk = seed_kmer + m.seed_off, *pg = seed_gpos + m.seed_off, *pc = seed_ctg + m.seed_off;
    uint32_t *ok = skmer + m.seed_off, *og = sgpos + m.seed_off, *oc = sctg + m.seed_off;
    uint32_t *boff = boff_all + m.bucket_off;
    const uint32_t *rg = rec_goff + m.rec_goff_off;

    for (uint32_t b = tid; b < nb; b += 256) cnt[b] = 0;
    if (tid == 0) s_distinct = 0;
    __syncthreads();
    for (uint32_t s = tid; s < n; s += 256) atomicAdd(&cnt[kmer_bucket(pk[s] & SK_SEED_MASK, bits)], 1u);
    __syncthreads();
    // exclusive scan of the bucket counts -> boff (global); counters are reset for the scatter cursor
    uint32_t running = 0;
    for (uint32_t base = 0; base < nb; base += 256) {
        uint32_t b = base + tid;
        uint32_t v = b < nb ? cnt[b] : 0u;
        uint32_t total;
        uint32_t ex = block_excl_scan_256(v, wsum, total);
        if (b < nb) { boff[b] = running + ex; cnt[b] = 0; }
        running += total;
    }
    if (tid == 0) boff[nb] = n;
    __syncthreads();
    for (uint32_t s = tid; s < n; s += 256) {
        uint32_t km = pk[s];
        uint32_t b = kmer_bucket(km & SK_SEED_MASK, bits);
        uint32_t pos = boff[b] + atomicAdd(&cnt[b], 1u);
        ok[pos] = km; og[pos] = pg[s]; oc[pos] = pc[s];
    }
    __syncthreads();   // global writes of this workgroup are visible to it after the barrier
    // histogram of multiplicities reuses the counter array
    for (uint32_t b = tid; b < IDX_REP_HIST; b += 256) cnt[b] = 0;
    __syncthreads();
    uint32_t my_distinct = 0;
    for (uint32_t b = tid; b < nb; b += 256) {
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
    for (uint32_t base = 0; base < n; base += 256) {
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
        uint32_t ex = block_excl_scan_256(flag, wsum, total);
        if (s < n) {
            pchunk[m.seed_off + s] = crun + ex + flag - 1u;
            if (flag) chunk_start_all[m.chunk_off + crun + ex] = s;
        }
        crun += total;
    }
    if (tid == 0) { meta[g].n_chunks = crun; chunk_start_all[m.chunk_off + crun] = n; }
}

// record look-up table: lut[b] = index of the kept record that holds genome position b << REC_LUT_SHIFT
// (records are >= 500 bp, so the wanted record is at most a few steps after lut[pos >> REC_LUT_SHIFT])
__global__ __launch_bounds__(256) void rec_lut_kernel(const GenomeMeta *__restrict__ meta, const uint32_t *__restrict__ rec_goff,
                                                     uint16_t *__restrict__ lut_all)
{
    const GenomeMeta m = meta[blockIdx.x];
    const uint32_t *rg = rec_goff + m.rec_goff_off;
    const uint32_t n = (uint32_t)(m.total_len >> REC_LUT_SHIFT) + 1u;
    uint16_t *lut = lut_all + m.rec_lut_off;
    for (uint32_t b = threadIdx.x; b < n; b += blockDim.x) {
        const uint32_t pos = b << REC_LUT_SHIFT;
        uint32_t lo = 0, hi = m.n_rec;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (rg[mid] <= pos) lo = mid; else hi = mid;
        }
        lut[b] = (uint16_t)(lo < 65535u ? lo : 65535u);
    }
}

void index_impl(skder_sketches *s)
{
    if (s->indexed) return;
    skder_ctx *ctx = s->ctx;
    hipStream_t st = ctx->stream;
    const uint32_t G = s->n_genomes;
    s->h_meta.resize(G);
    uint64_t boff_total = 0, rg = 0, chunk_total = 0, lut_total = 0;
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
        m.rec_lut_off = lut_total;
        lut_total += (m.total_len >> REC_LUT_SHIFT) + 1;
        chunk_total += m.total_len / ANI_CHUNK_LEN + m.n_rec + 2;   // upper bound on chunks + sentinel
    }
    const uint64_t ns = s->h_seed_off[G];
    s->d_meta.resize(G, st);
    s->d_rec_goff.resize(s->h_rec_goff.size() + 1, st);
    s->skmer.resize(ns + 1, st); s->sgpos.resize(ns + 1, st); s->sctg.resize(ns + 1, st);
    s->pchunk.resize(ns + 1, st);
    s->boff.resize(boff_total + 1, st);
    s->chunk_start.resize(chunk_total + 1, st);
    s->rec_lut.resize(lut_total + 1, st);
    if (G) {
        HIPCHECK(hipMemcpyAsync(s->d_meta.p, s->h_meta.data(), G * sizeof(GenomeMeta), hipMemcpyHostToDevice, st));
        HIPCHECK(hipMemcpyAsync(s->d_rec_goff.p, s->h_rec_goff.data(), s->h_rec_goff.size() * 4, hipMemcpyHostToDevice, st));
        HIPCHECK(hipEventRecord(ctx->ev[3], st));
        uint32_t max_bits = 12;   // the multiplicity histogram needs 4096 counters
        for (uint32_t g = 0; g < G; g++) max_bits = s->h_meta[g].bucket_bits > max_bits ? s->h_meta[g].bucket_bits : max_bits;
        hipLaunchKernelGGL(index_genome_kernel, dim3(G), dim3(256), (1u << max_bits) * 4, st, s->d_meta.p,
                           s->d_rec_goff.p, s->seed_kmer.p, s->seed_gpos.p, s->seed_ctg.p, s->skmer.p, s->sgpos.p,
                           s->sctg.p, s->boff.p, s->pchunk.p, s->chunk_start.p);
        hipLaunchKernelGGL(rec_lut_kernel, dim3(G), dim3(256), 0, st, s->d_meta.p, s->d_rec_goff.p, s->rec_lut.p);
        HIPCHECK(hipEventRecord(ctx->ev[4], st));
        HIPCHECK(hipMemcpyAsync(s->h_meta.data(), s->d_meta.p, G * sizeof(GenomeMeta), hipMemcpyDeviceToHost, st));
        HIPCHECK(hipStreamSynchronize(st));
        float ms = 0;
        HIPCHECK(hipEventElapsedTime(&ms, ctx->ev[3], ctx->ev[4]));
        ctx->timing[1] += ms;
    }
    s->indexed = true;
}
