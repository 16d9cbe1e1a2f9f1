// chain.hip -- anchors, chunked chaining, ANI / aligned fraction for a list of genome pairs.
//
// Device restatement of oracle/ani_oracle.c oracle_pair() (steps 1-6); integer results are
// bit-identical by construction, the few double operations are + - * / in the oracle's order
// (this file is compiled with -ffp-contract=off).
//
//   anchors_kernel   one workgroup per pair.  The chunked genome's seeds are read in position order
//                    (coalesced), every seed probes the other genome's k-mer bucket index; matches
//                    are written in order through a workgroup scan: 12 B per anchor.
//   chain_kernel     one LANE per 20 kb chunk: banded DP (band 50, exact early exit on the running
//                    maximum), best-first chain extraction with back-tracking; emits chain records.
//   finalize_kernel  one workgroup per pair: chains into LDS, better-chain overlap filter as a
//                    parallel fix-point, fixed-point containment ANI, aligned fraction.
#include "device_utils.h"
#include "engine.h"
#include "screen.h"

struct SetView {
    const GenomeMeta *meta;
    const uint32_t *pkmer, *pgpos, *pchunk;   // position order
    const uint32_t *skmer, *sgpos, *sctg;     // bucket order
    const uint32_t *boff;
};

struct PairDesc {
    uint32_t q, r;          // chunked genome, other genome (indices inside their sets)
    uint32_t a_base, a_cap; // anchor region of this pair in the batch buffers
    uint32_t chunk_base;    // first entry of this pair in chunk_aoff (n_chunks + 1 entries)
    uint32_t n_chunks;
    uint32_t c_base, c_cap; // chain-record region
    uint32_t flags;         // bit0: chunked genome is the pair's Query; bit1: q in set B; bit2: r in set B
    uint32_t pad[3];
};

struct ChainRec {
    int32_t score;
    uint32_t n, n_seeds, q0, q1, r0, r1, rctg;
};

struct PairOut {
    int64_t fx_sum;
    uint64_t sum_seeds, sum_anchors, sum_span;
    uint32_t n_chains, n_chains_all, n_anchors, pad;
    double ani_raw, ani, af_q, af_r;   // q = chunked genome
};

#define USED_BIT 0x80000000u
#define FIN_LDS_CHAINS 2048

__global__ __launch_bounds__(256) void anchors_kernel(SetView A, SetView B, const PairDesc *__restrict__ pairs,
                                                      uint32_t *__restrict__ a_qi, uint32_t *__restrict__ a_r,
                                                      uint32_t *__restrict__ a_rctg, uint32_t *__restrict__ chunk_aoff,
                                                      uint32_t *__restrict__ pair_na, uint32_t *__restrict__ flags)
{
    __shared__ uint32_t wsum[4];
    const PairDesc pd = pairs[blockIdx.x];
    const SetView &QS = (pd.flags & 2u) ? B : A;
    const SetView &RS = (pd.flags & 4u) ? B : A;
    const GenomeMeta Q = QS.meta[pd.q], R = RS.meta[pd.r];
    const uint32_t *qk = QS.pkmer + Q.seed_off, *qc = QS.pchunk + Q.seed_off;
    const uint32_t *rk = RS.skmer + R.seed_off, *rg = RS.sgpos + R.seed_off, *rc = RS.sctg + R.seed_off;
    const uint32_t *rb = RS.boff + R.bucket_off;
    const uint32_t *qsk = QS.skmer + Q.seed_off, *qb = QS.boff + Q.bucket_off;
    const uint32_t tid = threadIdx.x, nq = Q.n_seeds;
    uint32_t running = 0;
    for (uint32_t base = 0; base < nq; base += 256) {
        const uint32_t s = base + tid;
        uint32_t cnt = 0, first = 0, km = 0;
        if (s < nq) {
            km = qk[s];
            const uint32_t kmer = km & SK_SEED_MASK;
            const uint32_t b = kmer_bucket(kmer, R.bucket_bits);
            const uint32_t lo = rb[b], hi = rb[b + 1];
            for (uint32_t e = lo; e < hi; e++) {
                const uint32_t k2 = rk[e] & SK_SEED_MASK;
                if (k2 == kmer) { if (!cnt) first = e; cnt++; }
                else if (k2 > kmer) break;
            }
            if (cnt > R.rep_cut) cnt = 0;
            if (cnt && Q.rep_cut != 0xFFFFFFFFu) {   // multiplicity inside the chunked genome itself
                const uint32_t b2 = kmer_bucket(kmer, Q.bucket_bits);
                uint32_t m2 = 0;
                for (uint32_t e = qb[b2]; e < qb[b2 + 1]; e++) m2 += ((qsk[e] & SK_SEED_MASK) == kmer);
                if (m2 > Q.rep_cut) cnt = 0;
            }
        }
        uint32_t total;
        const uint32_t ex = block_excl_scan_256(cnt, wsum, total);
        const uint32_t at = running + ex;
        if (s < nq) {
            const uint32_t ck = qc[s];
            if (s == 0 || qc[s - 1] != ck) chunk_aoff[pd.chunk_base + ck] = pd.a_base + (at < pd.a_cap ? at : pd.a_cap);
        }
        if (cnt) {
            if (at + cnt <= pd.a_cap) {
                for (uint32_t t = 0; t < cnt; t++) {
                    const uint32_t idx = pd.a_base + at + t;
                    const uint32_t rkm = rk[first + t];
                    const uint32_t rev = (km >> 31) != (rkm >> 31);
                    a_qi[idx] = s;
                    a_r[idx] = rg[first + t] | (rev ? USED_BIT : 0u);
                    a_rctg[idx] = rc[first + t];
                }
            } else {
                atomicOr(&flags[0], 4u);
            }
        }
        running += total;
    }
    if (tid == 0) {
        chunk_aoff[pd.chunk_base + pd.n_chunks] = pd.a_base + (running < pd.a_cap ? running : pd.a_cap);
        pair_na[blockIdx.x] = running;
    }
}

__global__ __launch_bounds__(256) void chain_kernel(SetView A, SetView B, const PairDesc *__restrict__ pairs, uint32_t npairs,
                                                    uint32_t total_entries, const uint32_t *__restrict__ a_qi,
                                                    const uint32_t *__restrict__ a_r, const uint32_t *__restrict__ a_rctg,
                                                    const uint32_t *__restrict__ chunk_aoff, int32_t *__restrict__ F,
                                                    uint32_t *__restrict__ BP, ChainRec *__restrict__ chains,
                                                    uint32_t *__restrict__ pair_nch, uint32_t *__restrict__ flags)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= total_entries) return;
    uint32_t lo = 0, hi = npairs;
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (pairs[mid].chunk_base <= t) lo = mid; else hi = mid;
    }
    const PairDesc pd = pairs[lo];
    if (t - pd.chunk_base >= pd.n_chunks) return;   // the end sentinel
    const uint32_t a0 = chunk_aoff[t], a1 = chunk_aoff[t + 1];
    if (a1 <= a0) return;
    const uint32_t n = a1 - a0;
    const SetView &QS = (pd.flags & 2u) ? B : A;
    const uint32_t *qg = QS.pgpos + QS.meta[pd.q].seed_off;
    const uint32_t *qi = a_qi + a0, *ar = a_r + a0, *ac = a_rctg + a0;
    int32_t *f = F + a0;
    uint32_t *bp = BP + a0;

    // banded chaining; `runmax` bounds every f[j] seen so far, so once best >= runmax + score no
    // remaining predecessor can beat it (exact early exit, same result as the full band)
    int32_t runmax = -0x40000000;
    for (uint32_t i = 0; i < n; i++) {
        const int32_t qp = (int32_t)qg[qi[i]];
        const uint32_t rr = ar[i], rc = ac[i];
        const int32_t rp = (int32_t)(rr & 0x7FFFFFFFu);
        const uint32_t rev = rr >> 31;
        int32_t best = ANI_ANCHOR_SCORE, bj = -1;
        const uint32_t jlo = i > ANI_BAND ? i - ANI_BAND : 0u;
        for (uint32_t j = i; j-- > jlo;) {
            if (best >= runmax + ANI_ANCHOR_SCORE) break;
            const int32_t dq = qp - (int32_t)qg[qi[j]];
            if (dq > ANI_BP_BAND) break;
            const uint32_t rj = ar[j];
            if (ac[j] != rc || (rj >> 31) != rev) continue;
            const int32_t rpj = (int32_t)(rj & 0x7FFFFFFFu);
            const int32_t dr = rev ? rpj - rp : rp - rpj;
            if (dq <= 0 || dr <= 0) continue;
            if (dq > ANI_MAX_LIN || dr > ANI_MAX_LIN) continue;
            const int32_t gap = dq > dr ? dq - dr : dr - dq;
            if (gap > ANI_MAX_GAP) continue;
            const int32_t sc = f[j] + ANI_ANCHOR_SCORE - gap;
            if (sc > best) { best = sc; bj = (int32_t)j; }
        }
        f[i] = best;
        bp[i] = (uint32_t)(bj + 1);
        runmax = best > runmax ? best : runmax;
    }
    // chains: best end first (ties: lowest index); back-track until the start or a used anchor
    for (;;) {
        int32_t bestv = ANI_ANCHOR_SCORE, besti = -1;
        for (uint32_t i = 0; i < n; i++) {
            const int32_t v = f[i];
            if (v > bestv) { bestv = v; besti = (int32_t)i; }
        }
        if (besti < 0) break;
        uint32_t cnt = 0, rmin = 0xFFFFFFFFu, rmax = 0;
        int32_t cur = besti, first = besti;
        while (cur >= 0) {
            const uint32_t b = bp[cur];
            if (b & USED_BIT) break;
            cnt++;
            first = cur;
            const uint32_t rp = ar[cur] & 0x7FFFFFFFu;
            rmin = rp < rmin ? rp : rmin;
            rmax = rp > rmax ? rp : rmax;
            cur = (int32_t)(b & 0x7FFFFFFFu) - 1;
        }
        if (cnt < ANI_MIN_ANCHORS) { f[besti] = (int32_t)0x80000000; continue; }
        cur = besti;
        while (cur >= 0) {
            const uint32_t b = bp[cur];
            if (b & USED_BIT) break;
            bp[cur] = b | USED_BIT;
            f[cur] = (int32_t)0x80000000;
            cur = (int32_t)(b & 0x7FFFFFFFu) - 1;
        }
        const uint32_t slot = atomicAdd(&pair_nch[lo], 1u);
        if (slot < pd.c_cap) {
            ChainRec c;
            c.score = bestv;
            c.n = cnt;
            c.n_seeds = qi[besti] - qi[first] + 1;
            c.q0 = qg[qi[first]];
            c.q1 = qg[qi[besti]];
            c.r0 = rmin; c.r1 = rmax;
            c.rctg = ac[besti];
            chains[pd.c_base + slot] = c;
        } else {
            atomicOr(&flags[0], 8u);
        }
    }
}

// round(2^32 * (num/den)^(1/15)): Newton on doubles, + - * / only (oracle_root_fx)
__device__ __forceinline__ uint32_t root_fx(uint32_t num, uint32_t den)
{
    if (den == 0 || num == 0) return 0;
    if (num >= den) return 0xFFFFFFFFu;
    const double x = (double)num / (double)den;
    double y = 1.0;
    const double km1 = (double)(ANI_K - 1), kk = (double)ANI_K;
    for (int it = 0; it < ANI_ROOT_ITERS; it++) {
        double yp = 1.0;
#pragma unroll
        for (int i = 0; i < ANI_K - 1; i++) yp = yp * y;
        y = (km1 * y + x / yp) / kk;
    }
    const double s = y * ANI_FX_ONE + 0.5;
    if (s >= 4294967295.0) return 0xFFFFFFFFu;
    return (uint32_t)s;
}

__device__ __forceinline__ double calibrate_ani(double ani_raw)
{
    const double cx[ANI_CAL_N] = ANI_CAL_X;
    const double cy[ANI_CAL_N] = ANI_CAL_Y;
    double d = 100.0 * (1.0 - ani_raw);
    if (d < 0.0) d = 0.0;
    double out;
    if (d >= cx[ANI_CAL_N - 1]) {
        out = cy[ANI_CAL_N - 1] + (d - cx[ANI_CAL_N - 1]);
    } else {
        int i = 0;
        while (i + 2 < ANI_CAL_N && d >= cx[i + 1]) i++;
        const double t = (d - cx[i]) / (cx[i + 1] - cx[i]);
        out = cy[i] + t * (cy[i + 1] - cy[i]);
    }
    double a = 1.0 - out / 100.0;
    if (a < 0.0) a = 0.0;
    if (a > 1.0) a = 1.0;
    return a;
}

// is chain j ranked before chain i? (score desc, q0 asc, r0 asc, q1 asc) -- ani_oracle.c cmp_chain
__device__ __forceinline__ bool better(const int32_t *sc, const uint32_t *q0, const uint32_t *r0, const uint32_t *q1, uint32_t j, uint32_t i)
{
    if (sc[j] != sc[i]) return sc[j] > sc[i];
    if (q0[j] != q0[i]) return q0[j] < q0[i];
    if (r0[j] != r0[i]) return r0[j] < r0[i];
    return q1[j] < q1[i];
}

__global__ __launch_bounds__(256) void finalize_kernel(SetView A, SetView B, const PairDesc *__restrict__ pairs,
                                                       const ChainRec *__restrict__ chains, const uint32_t *__restrict__ pair_nch,
                                                       const uint32_t *__restrict__ pair_na, PairOut *__restrict__ out,
                                                       uint32_t *__restrict__ flags)
{
    __shared__ int32_t sc[FIN_LDS_CHAINS];
    __shared__ uint32_t q0[FIN_LDS_CHAINS], q1[FIN_LDS_CHAINS], r0[FIN_LDS_CHAINS], r1[FIN_LDS_CHAINS], rc[FIN_LDS_CHAINS];
    __shared__ uint32_t na[FIN_LDS_CHAINS], nsd[FIN_LDS_CHAINS];
    __shared__ uint8_t state[FIN_LDS_CHAINS];   // 0 unknown, 1 kept, 2 dropped
    __shared__ unsigned long long s_fx, s_seeds, s_anch, s_span;
    __shared__ uint32_t s_kept, s_unknown;

    const PairDesc pd = pairs[blockIdx.x];
    const uint32_t tid = threadIdx.x;
    uint32_t n = pair_nch[blockIdx.x];
    if (n > pd.c_cap) n = pd.c_cap;
    if (n > FIN_LDS_CHAINS) {
        if (tid == 0) atomicOr(&flags[0], 16u);
        n = FIN_LDS_CHAINS;
    }
    if (tid == 0) { s_fx = 0; s_seeds = 0; s_anch = 0; s_span = 0; s_kept = 0; s_unknown = 0; }
    for (uint32_t i = tid; i < n; i += 256) {
        const ChainRec c = chains[pd.c_base + i];
        sc[i] = c.score; q0[i] = c.q0; q1[i] = c.q1; r0[i] = c.r0; r1[i] = c.r1; rc[i] = c.rctg;
        na[i] = c.n; nsd[i] = c.n_seeds;
        state[i] = 0;
    }
    __syncthreads();
    // a chain is dropped when ONE better kept chain on the same record covers more than half of
    // its span on the other genome.  Chains without any better overlapping chain are kept at once;
    // the rest resolve in rounds, each chain waiting for its better overlapping chains.
    for (;;) {
        uint32_t my_unknown = 0;
        for (uint32_t i = tid; i < n; i += 256) {
            if (state[i]) continue;
            const uint32_t li = r1[i] - r0[i];
            bool dropped = false, pending = false;
            for (uint32_t j = 0; j < n; j++) {
                if (rc[j] != rc[i] || j == i) continue;
                const uint32_t lo = r0[i] > r0[j] ? r0[i] : r0[j];
                const uint32_t hi = r1[i] < r1[j] ? r1[i] : r1[j];
                if (hi <= lo) continue;
                if ((uint64_t)ANI_REF_OVERLAP_DEN * (hi - lo) <= (uint64_t)ANI_REF_OVERLAP_NUM * li) continue;
                if (!better(sc, q0, r0, q1, j, i)) continue;
                const uint8_t sj = state[j];
                if (sj == 1) { dropped = true; break; }
                if (sj == 0) pending = true;
            }
            if (dropped) state[i] = 2;
            else if (!pending) state[i] = 1;
            else my_unknown++;
        }
        if (my_unknown) atomicAdd(&s_unknown, my_unknown);
        __syncthreads();
        const uint32_t u = s_unknown;
        __syncthreads();
        if (tid == 0) s_unknown = 0;
        __syncthreads();
        if (!u) break;
    }
    unsigned long long fx = 0, sd = 0, an = 0, sp = 0;
    uint32_t kept = 0;
    for (uint32_t i = tid; i < n; i += 256) {
        if (state[i] != 1) continue;
        fx += (unsigned long long)nsd[i] * (unsigned long long)root_fx(na[i], nsd[i]);
        sd += nsd[i];
        an += na[i];
        sp += q1[i] - q0[i];
        kept++;
    }
    if (kept) {
        atomicAdd(&s_fx, fx); atomicAdd(&s_seeds, sd); atomicAdd(&s_anch, an); atomicAdd(&s_span, sp);
        atomicAdd(&s_kept, kept);
    }
    __syncthreads();
    if (tid == 0) {
        const SetView &QS = (pd.flags & 2u) ? B : A;
        const SetView &RS = (pd.flags & 4u) ? B : A;
        PairOut o;
        o.fx_sum = (int64_t)s_fx; o.sum_seeds = s_seeds; o.sum_anchors = s_anch; o.sum_span = s_span;
        o.n_chains = s_kept; o.n_chains_all = n; o.n_anchors = pair_na[blockIdx.x]; o.pad = 0;
        o.ani_raw = 0.0; o.ani = 0.0;
        if (s_seeds) {
            o.ani_raw = ((double)o.fx_sum / (double)s_seeds) / ANI_FX_ONE;
            o.ani = calibrate_ani(o.ani_raw);
        }
        const double Bv = (double)(s_span + (unsigned long long)ANI_PAD * s_kept);
        const uint64_t tq = QS.meta[pd.q].total_len, tr = RS.meta[pd.r].total_len;
        double afq = tq ? Bv / (double)tq : 0.0, afr = tr ? Bv / (double)tr : 0.0;
        if (afq > 1.0) afq = 1.0;
        if (afr > 1.0) afr = 1.0;
        o.af_q = afq; o.af_r = afr;
        out[blockIdx.x] = o;
    }
}

// ---------------------------------------------------------------------------------------------
// host orchestration

static SetView view_of(skder_sketches *s)
{
    SetView v;
    v.meta = s->d_meta.p;
    v.pkmer = s->seed_kmer.p; v.pgpos = s->seed_gpos.p; v.pchunk = s->pchunk.p;
    v.skmer = s->skmer.p; v.sgpos = s->sgpos.p; v.sctg = s->sctg.p; v.boff = s->boff.p;
    return v;
}

// which genome is chunked: smaller T*(T/n_records); ties chunk the query (ani_oracle.c chunk_query)
static bool chunk_the_query(const GenomeMeta &ref, const GenomeMeta &query)
{
    double tq = (double)query.total_len, tr = (double)ref.total_len;
    double sq = tq * (tq / (double)(query.n_rec ? query.n_rec : 1));
    double sr = tr * (tr / (double)(ref.n_rec ? ref.n_rec : 1));
    return sq <= sr;
}

// pairs: (ref genome in set A, query genome in set B); for the triangle A == B.
void chain_pairs(skder_sketches *SA, skder_sketches *SB, const std::vector<uint32_t> &pref, const std::vector<uint32_t> &pquery,
                 std::vector<skder_edge_t> &edges)
{
    skder_ctx *ctx = SA->ctx;
    hipStream_t st = ctx->stream;
    const size_t np = pref.size();
    size_t budget = 256u << 20;   // anchors per batch (x 20 B)
    if (const char *e = getenv("SKDER_AMD_ANCHOR_BUDGET")) budget = strtoull(e, nullptr, 10);
    DevBuf<PairDesc> d_pairs;
    DevBuf<uint32_t> a_qi, a_r, a_rctg, BP, chunk_aoff, pair_na, pair_nch;
    DevBuf<int32_t> F;
    DevBuf<ChainRec> chains;
    DevBuf<PairOut> d_out;
    std::vector<PairDesc> hp;
    std::vector<PairOut> ho;
    const SetView VA = view_of(SA), VB = view_of(SB);
    double t_anchor = 0, t_chain = 0, t_fin = 0;
    uint64_t tot_anchors = 0;
    size_t p0 = 0;
    while (p0 < np) {
        hp.clear();
        uint64_t acap = 0, nentries = 0, ccap = 0;
        size_t p = p0;
        for (; p < np; p++) {
            const GenomeMeta &mr = SA->h_meta[pref[p]], &mq = SB->h_meta[pquery[p]];
            const bool cq = chunk_the_query(mr, mq);
            const GenomeMeta &Q = cq ? mq : mr;
            PairDesc d;
            memset(&d, 0, sizeof d);
            d.q = cq ? pquery[p] : pref[p];
            d.r = cq ? pref[p] : pquery[p];
            d.flags = (cq ? 1u : 0u) | (cq ? 2u : 0u) | (cq ? 0u : 4u);   // Q in B iff cq; R in B iff !cq
            d.a_cap = 2u * Q.n_seeds + 1024u;
            d.n_chunks = Q.n_chunks;
            d.c_cap = 4u * Q.n_chunks + 64u;
            if (!hp.empty() && acap + d.a_cap > budget) break;
            if (acap + d.a_cap > 0xFFFF0000ull || nentries + d.n_chunks + 1 > 0xFFFF0000ull) break;
            d.a_base = (uint32_t)acap; d.chunk_base = (uint32_t)nentries; d.c_base = (uint32_t)ccap;
            acap += d.a_cap; nentries += d.n_chunks + 1; ccap += d.c_cap;
            hp.push_back(d);
        }
        const uint32_t nb = (uint32_t)hp.size();
        d_pairs.resize(nb, st);
        a_qi.resize(acap + 1, st); a_r.resize(acap + 1, st); a_rctg.resize(acap + 1, st);
        F.resize(acap + 1, st); BP.resize(acap + 1, st);
        chunk_aoff.resize(nentries + 1, st);
        pair_na.resize(nb, st); pair_nch.resize(nb, st);
        chains.resize(ccap + 1, st);
        d_out.resize(nb, st);
        HIPCHECK(hipMemcpyAsync(d_pairs.p, hp.data(), nb * sizeof(PairDesc), hipMemcpyHostToDevice, st));
        HIPCHECK(hipMemsetAsync(pair_nch.p, 0, nb * 4, st));
        HIPCHECK(hipMemsetAsync(ctx->d_flags, 0, 64, st));
        HIPCHECK(hipEventRecord(ctx->ev[5], st));
        hipLaunchKernelGGL(anchors_kernel, dim3(nb), dim3(256), 0, st, VA, VB, d_pairs.p, a_qi.p, a_r.p, a_rctg.p,
                           chunk_aoff.p, pair_na.p, ctx->d_flags);
        HIPCHECK(hipEventRecord(ctx->ev[6], st));
        hipLaunchKernelGGL(chain_kernel, dim3((unsigned)((nentries + 255) / 256)), dim3(256), 0, st, VA, VB, d_pairs.p, nb,
                           (uint32_t)nentries, a_qi.p, a_r.p, a_rctg.p, chunk_aoff.p, F.p, BP.p, chains.p, pair_nch.p,
                           ctx->d_flags);
        HIPCHECK(hipEventRecord(ctx->ev[7], st));
        hipLaunchKernelGGL(finalize_kernel, dim3(nb), dim3(256), 0, st, VA, VB, d_pairs.p, chains.p, pair_nch.p, pair_na.p,
                           d_out.p, ctx->d_flags);
        HIPCHECK(hipEventRecord(ctx->ev[8], st));
        ho.resize(nb);
        uint32_t h_flags = 0;
        HIPCHECK(hipMemcpyAsync(ho.data(), d_out.p, nb * sizeof(PairOut), hipMemcpyDeviceToHost, st));
        HIPCHECK(hipMemcpyAsync(&h_flags, ctx->d_flags, 4, hipMemcpyDeviceToHost, st));
        HIPCHECK(hipStreamSynchronize(st));
        if (h_flags & 4u) throw SkError("anchor buffer overflow (a pair has more than 2*seeds+1024 anchors)");
        if (h_flags & 8u) throw SkError("chain buffer overflow (a pair has more than 4*chunks+64 chains)");
        if (h_flags & 16u) throw SkError("pair with more than 2048 chains is not supported");
        float ms;
        HIPCHECK(hipEventElapsedTime(&ms, ctx->ev[5], ctx->ev[6])); t_anchor += ms;
        HIPCHECK(hipEventElapsedTime(&ms, ctx->ev[6], ctx->ev[7])); t_chain += ms;
        HIPCHECK(hipEventElapsedTime(&ms, ctx->ev[7], ctx->ev[8])); t_fin += ms;
        for (uint32_t i = 0; i < nb; i++) {
            const PairOut &o = ho[i];
            tot_anchors += o.n_anchors;
            if (!o.n_chains || !(o.ani > 0.0)) continue;
            const bool cq = hp[i].flags & 1u;
            skder_edge_t e;
            e.ref = pref[p0 + i]; e.query = pquery[p0 + i];
            e.ani = o.ani;
            e.af_query = cq ? o.af_q : o.af_r;
            e.af_ref = cq ? o.af_r : o.af_q;
            e.n_chains = o.n_chains; e.n_anchors = o.n_anchors;
            e.aligned_bases = o.sum_span + (uint64_t)ANI_PAD * o.n_chains;
            e.ani_fx_sum = o.fx_sum; e.sum_seeds = o.sum_seeds;
            edges.push_back(e);
        }
        p0 = p;
    }
    ctx->timing[3] = t_anchor; ctx->timing[4] = t_chain; ctx->timing[5] = t_fin;
    ctx->timing[6] = (double)np; ctx->timing[7] = (double)tot_anchors;
}

void triangle_rows_impl(skder_sketches *s, uint32_t row_begin, uint32_t row_stride, double screen_pct)
{
    skder_ctx *ctx = s->ctx;
    if (!s->indexed) index_impl(s);
    ctx->edges.clear();
    std::vector<uint32_t> rows;
    for (uint32_t i = row_begin; i < s->n_genomes; i += (row_stride ? row_stride : 1)) rows.push_back(i);
    std::vector<uint32_t> prow, ppart;
    HIPCHECK(hipEventRecord(ctx->ev[9], ctx->stream));
    screen_pairs(s, s, rows, true, screen_pct, prow, ppart);
    HIPCHECK(hipEventRecord(ctx->ev[10], ctx->stream));
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    float ms;
    HIPCHECK(hipEventElapsedTime(&ms, ctx->ev[9], ctx->ev[10]));
    ctx->timing[2] = ms;
    // triangle row (i, j): Ref = i, Query = j
    chain_pairs(s, s, prow, ppart, ctx->edges);
}

void rectangle_impl(skder_sketches *refs, skder_sketches *queries, double screen_pct)
{
    skder_ctx *ctx = refs->ctx;
    if (!refs->indexed) index_impl(refs);
    if (!queries->indexed) index_impl(queries);
    ctx->edges.clear();
    std::vector<uint32_t> rows(queries->n_genomes);
    for (uint32_t i = 0; i < queries->n_genomes; i++) rows[i] = i;
    std::vector<uint32_t> prow, ppart;
    HIPCHECK(hipEventRecord(ctx->ev[9], ctx->stream));
    screen_pairs(refs, queries, rows, false, screen_pct, prow, ppart);
    HIPCHECK(hipEventRecord(ctx->ev[10], ctx->stream));
    HIPCHECK(hipStreamSynchronize(ctx->stream));
    float ms;
    HIPCHECK(hipEventElapsedTime(&ms, ctx->ev[9], ctx->ev[10]));
    ctx->timing[2] = ms;
    // rows are queries, partners are references
    chain_pairs(refs, queries, ppart, prow, ctx->edges);
}
