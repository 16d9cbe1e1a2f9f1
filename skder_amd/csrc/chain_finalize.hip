// chain_finalize.hip -- one workgroup per pair: its chains into LDS, the better-chain overlap filter as a parallel fix-point,
// the sums over the kept chains, the two 15-th roots, the ANI model and the aligned fractions (oracle_pair() steps 5-6)
#include "chain.h"

// (num/den)^(1/15): Newton on doubles, + - * / only (ani_oracle.c oracle_root)
__device__ __forceinline__ double root_k(uint64_t num, uint64_t den)
{
    if (den == 0 || num == 0) return 0.0;
    if (num >= den) return 1.0;
    const double x = (double)num / (double)den;
    double y = 1.0;
    const double km1 = (double)(ANI_K - 1), kk = (double)ANI_K;
    for (int it = 0; it < ANI_ROOT_ITERS; it++) {
        double yp = 1.0;
#pragma unroll
        for (int i = 0; i < ANI_K - 1; i++) yp = yp * y;
        const double yn = (km1 * y + x / yp) / kk;
        if (yn == y) break;      // a fixed point: every further iteration returns the same value (the result is unchanged)
        y = yn;
    }
    return y;
}

// the two-estimate ANI model of include/skder_amd_spec.h (ani_oracle.c oracle_model_ani)
__device__ __forceinline__ double model_ani(double ani_cell, double ani_span)
{
    const double d = ANI_CAL_CELL * (100.0 * (1.0 - ani_cell)) + ANI_CAL_SPAN * (100.0 * (1.0 - ani_span));
    double a = 1.0 - d / 100.0;
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
    if (q1[j] != q1[i]) return q1[j] < q1[i];
    // chains equal in every key (repeats can yield two chains with the same ends and score): the sort of the oracle
    // puts one of them first, and that one drops the other; which one cannot matter, so the array order decides
    return j < i;
}

// GLOBAL = false: the chain arrays of the pair live in dynamic LDS (`cap_arg` chains, sized per batch by the
// host), one workgroup per pair of the batch.  GLOBAL = true: the same code for the few pairs with more
// chains than LDS holds (repeat-rich genomes): arrays in a global workspace, workgroup b handles pair
// glist[b] with capacity gcap[b] at gws + goff[b].
// the most chunks of a pair whose marks live in LDS (the bin counters' words); a test build lowers it to send every pair through the global marks
#ifndef FIN_THREADS
#define FIN_THREADS 192     // threads of the workgroup of a pair whose chains fit in LDS: ten pairs per CU (with the LDS sized to the batch, chain.hip);
                            // measured 128 / 192 / 256: 423 / 386 / 500 us per launch on the headline, 0.374 / 0.367 / 0.365 us per pair on real structure
#endif
#ifndef FIN_LDS_MARKS
#define FIN_LDS_MARKS (FIN_BINS + 1u)
#endif
static_assert(FIN_LDS_MARKS <= FIN_BINS + 1u, "the marks reuse the bin counters");

#ifdef FIN_TIMING
// measurement build: wall-clock ticks (100 MHz) of the phases of a workgroup, summed over the workgroups
__device__ unsigned long long fin_ticks[8];
#define FIN_T(k) do { if (tid == 0) { const unsigned long long now_ = wall_clock64(); atomicAdd(&fin_ticks[k], now_ - t_prev_); t_prev_ = now_; } } while (0)
void finalize_timing_dump()
{
    unsigned long long h[8];
    HIPCHECK(hipMemcpyFromSymbol(h, HIP_SYMBOL(fin_ticks), sizeof h));
    fprintf(stderr, "[skder_amd] finalize ticks (10 ns): gather %llu, bins %llu, rounds %llu, sums %llu, roots+out %llu; workgroups %llu\n", h[0], h[1], h[2], h[3], h[4], h[7]);
    memset(h, 0, sizeof h);
    HIPCHECK(hipMemcpyToSymbol(HIP_SYMBOL(fin_ticks), h, sizeof h));
}
#else
#define FIN_T(k) do { } while (0)
#endif

template <bool GLOBAL, int NT>
__global__ __launch_bounds__(NT) void finalize_kernel_t(SetView A, SetView B, const PairDesc *__restrict__ pairs,
                                                         const ChainRec *__restrict__ fast_chains, uint32_t fast_stride, const uint32_t *__restrict__ chunk_state,
                                                         const ChainRec *__restrict__ chains, const uint32_t *__restrict__ pair_nch,
                                                         const uint32_t *__restrict__ pair_na, PairOut *__restrict__ out,
                                                         uint32_t *__restrict__ flags, uint32_t *__restrict__ chunk_mark, uint32_t cap_arg,
                                                         unsigned char *__restrict__ gws, const uint64_t *__restrict__ goff,
                                                         const uint32_t *__restrict__ glist, const uint32_t *__restrict__ gcap)
{
    // 8 word arrays + 1 byte array + 1 u16 array of `lds_cap` chains
    extern __shared__ __attribute__((aligned(16))) unsigned char fin_smem[];
    const uint32_t pidx = GLOBAL ? glist[blockIdx.x] : blockIdx.x;
    const uint32_t lds_cap = GLOBAL ? gcap[blockIdx.x] : cap_arg;
    unsigned char *const arrays = GLOBAL ? gws + goff[blockIdx.x] : fin_smem;
    int32_t *sc = reinterpret_cast<int32_t *>(arrays);
    uint32_t *q0 = reinterpret_cast<uint32_t *>(arrays) + lds_cap, *q1 = q0 + lds_cap, *r0 = q1 + lds_cap, *r1 = r0 + lds_cap;
    uint32_t *ckc = r1 + lds_cap, *na = ckc + lds_cap, *nsd = na + lds_cap;   // ckc: the chain's chunk inside the pair
    uint8_t *state = reinterpret_cast<uint8_t *>(nsd + lds_cap);   // 0 unknown, 1 kept, 2 dropped
    uint16_t *order = reinterpret_cast<uint16_t *>(state + lds_cap);  // chain indices grouped by bin
    __shared__ unsigned long long s_cells, s_seeds, s_anch, s_span;
    __shared__ uint32_t s_kept, s_n;
    __shared__ uint32_t s_maxlen, s_maxr;
    __shared__ uint32_t bin_start[FIN_BINS + 2], bin_fill[FIN_BINS + 1];
    __shared__ uint32_t wsum[4];

    const PairDesc pd = pairs[pidx];
    const uint32_t tid = threadIdx.x;
#ifdef FIN_TIMING
    unsigned long long t_prev_ = wall_clock64();
    if (tid == 0) atomicAdd(&fin_ticks[7], 1ull);
#endif
    if (tid == 0) { s_cells = 0; s_seeds = 0; s_anch = 0; s_span = 0; s_kept = 0; s_n = 0; s_maxlen = 0; s_maxr = 0; }
    // what the last lane standing needs at the very end is requested now (the workgroup holds its LDS until then)
    const uint64_t len_q = ((pd.flags & 2u) ? B : A).meta[pd.q].total_len, len_r = ((pd.flags & 4u) ? B : A).meta[pd.r].total_len;
    const uint32_t n_anchors_pair = pair_na[pidx];
    // the seeds of this thread's chunk (the cell sum at the end wants them for the chunks that keep a chain): one trip for
    // a pair of up to 256 chunks, requested with the chunk states
    const uint32_t *const cst = ((pd.flags & 2u) ? B : A).chunk_start + pd.q_chunk_off;
    const uint32_t my_cell_seeds = tid < pd.n_chunks ? cst[tid + 1] - cst[tid] : 0u;
    __syncthreads();
    // gather: chains of the fast path (per-chunk slots) and of the slow path (per-pair list)
    uint32_t nslow = pair_nch[pidx];
    if (nslow > pd.c_cap) nslow = pd.c_cap;
    auto put = [&](const ChainRec &c) {
        const uint32_t d = atomicAdd(&s_n, 1u);       // (one atomic per wavefront and call, the lanes ranked by ballot, was measured: 3 % slower)
        if (d < lds_cap) {
            sc[d] = c.score; q0[d] = c.q0; q1[d] = c.q1; r0[d] = c.r0; r1[d] = c.r1; ckc[d] = c.chunk;
            na[d] = c.n; nsd[d] = c.n_seeds;
            state[d] = 0;
        }
    };
    // one chunk per thread: its state, then all of its chains at once (a 3 Mb genome has 150 chunks: one trip)
    for (uint32_t ck = tid; ck < pd.n_chunks; ck += NT) {
        const uint32_t st = chunk_state[pd.chunk_base + ck];
        if (st == CHUNK_SLOW || st == 0u) continue;
        // slot k of the chunk at fc[k * fast_stride]: slot-major, the first slots of a pair's chunks lie side by side (chain.h).
        // (The first two slots requested together with the state, before it is known whether they hold anything: no change.)
        const ChainRec *fc = fast_chains + (pd.chunk_base + ck);
        ChainRec c[3];        // the first three at once (nearly every chunk has no more), the rest one by one
#pragma unroll
        for (uint32_t k = 0; k < 3u; k++) if (k < st) c[k] = fc[(uint64_t)k * fast_stride];
#pragma unroll
        for (uint32_t k = 0; k < 3u; k++) if (k < st) put(c[k]);
        for (uint32_t k = 3u; k < st && k < FAST_SLOTS; k++) put(fc[(uint64_t)k * fast_stride]);
    }
    for (uint32_t i = tid; i < nslow; i += NT) put(chains[pd.c_base + i]);
    __syncthreads();
    uint32_t n = s_n;
    if (n > lds_cap) {
        // more chains than the arrays hold: report the number wanted (n_chains = ~0 marks the record) and
        // leave the pair to a launch with enough room
        if (tid == 0) {
            atomicOr(&flags[0], 16u);
            PairOut o;
            memset(&o, 0, sizeof o);
            o.n_chains = 0xFFFFFFFFu; o.n_chains_all = n;
            out[pidx] = o;
        }
        return;
    }
    FIN_T(0);
    // Spatial binning on the other genome so that a chain is compared only with chains that can
    // overlap it: bins of width 2^shift >= the longest chain, chains filed under the bin of r0; a chain
    // in bin b can only overlap chains of bins b-1, b, b+1.  (Exact for any input: a very long chain
    // just makes the bins wide.)
    {
        uint32_t ml = 0, mr = 0;
        for (uint32_t i = tid; i < n; i += NT) { ml = max(ml, r1[i] - r0[i]); mr = max(mr, r1[i]); }
        if (ml) atomicMax(&s_maxlen, ml);
        if (mr) atomicMax(&s_maxr, mr);
    }
    __syncthreads();
    uint32_t shift = 1;
    while ((1u << shift) <= s_maxlen) shift++;
    while ((s_maxr >> shift) >= FIN_BINS) shift++;
    const uint32_t nb_used = (s_maxr >> shift) + 1u;     // bins that can hold a chain (a 3 Mb genome with 20 kb chunks: ~90 of 1024)
    for (uint32_t b = tid; b <= nb_used; b += NT) bin_fill[b] = 0;
    __syncthreads();
    for (uint32_t i = tid; i < n; i += NT) atomicAdd(&bin_fill[r0[i] >> shift], 1u);
    __syncthreads();
    {
        uint32_t running = 0;
        for (uint32_t base = 0; base < nb_used; base += NT) {
            const uint32_t v = base + tid < nb_used ? bin_fill[base + tid] : 0u;
            uint32_t total;
            const uint32_t ex = block_excl_scan<NT / 64>(v, wsum, total);
            if (base + tid < nb_used) bin_start[base + tid] = running + ex;      // (the last trip is partial unless NT divides the bins in use)
            running += total;
        }
        __syncthreads();
        if (tid == 0) { bin_start[nb_used] = n; bin_start[nb_used + 1] = n; }
    }
    __syncthreads();
    for (uint32_t b = tid; b <= nb_used; b += NT) bin_fill[b] = 0;
    __syncthreads();
    for (uint32_t i = tid; i < n; i += NT) {
        const uint32_t b = r0[i] >> shift;
        order[bin_start[b] + atomicAdd(&bin_fill[b], 1u)] = (uint16_t)i;
    }
    __syncthreads();
    // the fill counters are free from here on: they become the marks of the chunks that keep a chain (a word per chunk;
    // a pair with more chunks than bins marks in global memory instead)
    FIN_T(1);
    const bool lds_marks = pd.n_chunks <= FIN_LDS_MARKS;
    uint32_t *const gmark = chunk_mark + pd.chunk_base;
    if (lds_marks) { for (uint32_t i = tid; i < pd.n_chunks; i += NT) bin_fill[i] = 0u; }
    else { for (uint32_t i = tid; i < pd.n_chunks; i += NT) gmark[i] = 0u; }
    // a chain is dropped when ONE better kept chain on the same record covers more than half of
    // its span on the other genome.  Chains without any better overlapping chain are kept at once;
    // the rest resolve in rounds, each chain waiting for its better overlapping chains.
    for (;;) {
        uint32_t my_unknown = 0;
        for (uint32_t i = tid; i < n; i += NT) {
            if (state[i]) continue;
            const uint32_t li = r1[i] - r0[i];
            bool dropped = false, pending = false;
            const uint32_t b = r0[i] >> shift;
            const uint32_t k0 = bin_start[b ? b - 1 : 0], k1 = bin_start[b + 2];    // bin_start[nb_used], [nb_used + 1] = n
            for (uint32_t k = k0; k < k1; k++) {
                const uint32_t j = order[k];
                if (j == i) continue;     // chains lie inside one record and positions are genome-linear: overlap implies the same record
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
        if (!__syncthreads_or((int)my_unknown)) break;
    }
    FIN_T(2);
    // sums over the kept chains; the cells (chunks) that hold one are marked (cleared before the rounds above, which end
    // on a barrier) and their seed counts summed afterwards
    unsigned long long sd = 0, an = 0, sp = 0, cs = 0;
    uint32_t kept = 0;
    for (uint32_t i = tid; i < n; i += NT) {
        if (state[i] != 1) continue;
        sd += nsd[i];
        an += na[i];
        sp += q1[i] - q0[i];
        kept++;
        if (lds_marks) bin_fill[ckc[i]] = 1u; else gmark[ckc[i]] = 1u;
    }
    __syncthreads();
    if (tid < pd.n_chunks && (lds_marks ? bin_fill[tid] : gmark[tid])) cs += my_cell_seeds;
    for (uint32_t i = tid + (uint32_t)NT; i < pd.n_chunks; i += NT)
        if (lds_marks ? bin_fill[i] : gmark[i]) cs += cst[i + 1] - cst[i];
    // wave-level reduction first: 4 LDS atomics per sum and workgroup instead of 256
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        cs += __shfl_down(cs, o, 64); sd += __shfl_down(sd, o, 64);
        an += __shfl_down(an, o, 64); sp += __shfl_down(sp, o, 64);
        kept += __shfl_down(kept, o, 64);
    }
    if ((tid & 63u) == 0 && (kept || cs)) {
        atomicAdd(&s_cells, cs); atomicAdd(&s_seeds, sd); atomicAdd(&s_anch, an); atomicAdd(&s_span, sp);
        atomicAdd(&s_kept, kept);
    }
    __syncthreads();
    FIN_T(3);
    if (tid >= 64) return;
    // the two 15-th roots are serial double arithmetic: one lane each
    double root = 0.0;
    if (tid < 2 && s_seeds) root = root_k(s_anch, tid == 0 ? s_cells : s_seeds);
    const double root_cell = __shfl(root, 0, 64), root_span = __shfl(root, 1, 64);
    if (tid == 0) {
        PairOut o;
        o.cell_seeds = s_cells; o.sum_seeds = s_seeds; o.sum_anchors = s_anch; o.sum_span = s_span;
        o.n_chains = s_kept; o.n_chains_all = n; o.n_anchors = n_anchors_pair; o.pad = 0;
        o.ani_raw = 0.0; o.ani_span = 0.0; o.ani = 0.0;
        if (s_seeds) {
            o.ani_raw = root_cell;
            o.ani_span = root_span;
            o.ani = model_ani(o.ani_raw, o.ani_span);
        }
        const double Bv = (double)(s_span + (unsigned long long)ANI_PAD * s_kept);
        const uint64_t tq = len_q, tr = len_r;
        double afq = tq ? Bv / (double)tq : 0.0, afr = tr ? Bv / (double)tr : 0.0;
        if (afq > 1.0) afq = 1.0;
        if (afr > 1.0) afr = 1.0;
        o.af_q = afq; o.af_r = afr;
        out[pidx] = o;
        FIN_T(4);
    }
}

void launch_finalize(hipStream_t st, unsigned grid, uint32_t lds_cap, SetView A, SetView B, const PairDesc *pairs, const ChainRec *fast_chains, uint32_t fast_stride,
                     const uint32_t *chunk_state, const ChainRec *chains, const uint32_t *pair_nch, const uint32_t *pair_na, PairOut *out,
                     uint32_t *flags, uint32_t *chunk_mark)
{
    hipLaunchKernelGGL((finalize_kernel_t<false, FIN_THREADS>), dim3(grid), dim3(FIN_THREADS), lds_cap * 35u, st, A, B, pairs, fast_chains, fast_stride, chunk_state, chains, pair_nch, pair_na,
                       out, flags, chunk_mark, lds_cap, nullptr, nullptr, nullptr, nullptr);
}
void launch_finalize_global(hipStream_t st, unsigned grid, SetView A, SetView B, const PairDesc *pairs, const ChainRec *fast_chains, uint32_t fast_stride,
                            const uint32_t *chunk_state, const ChainRec *chains, const uint32_t *pair_nch, const uint32_t *pair_na, PairOut *out,
                            uint32_t *flags, uint32_t *chunk_mark, unsigned char *gws, const uint64_t *goff, const uint32_t *glist, const uint32_t *gcap)
{
    hipLaunchKernelGGL((finalize_kernel_t<true, 256>), dim3(grid), dim3(256), 0, st, A, B, pairs, fast_chains, fast_stride, chunk_state, chains, pair_nch, pair_na, out, flags,
                       chunk_mark, 0u, gws, goff, glist, gcap);
}
void finalize_allow_large_lds() { HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&finalize_kernel_t<false, FIN_THREADS>), hipFuncAttributeMaxDynamicSharedMemorySize, 4096 * 35)); }
