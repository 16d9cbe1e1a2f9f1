// synth.h -- counter-based synthetic genomes: the base at every genome position is a
// pure function of (lineage seeds, position), so the device generator (synth_fill_kernel), numpy
// (skder_amd/synth.py) and any C host agree bit for bit.
//
//   species root      : iid ACGT from the species seed
//   strain ancestor   : 10 kb accessory segments replaced with strain-private sequence with
//                       probability acc_pct/100, plus substitutions at strain_sub_ppm
//   isolate           : substitutions at iso_sub_ppm
// genome_lineage[3g..] = {species_seed, strain_seed, isolate_seed}
// genome_params[4g..]  = {acc_pct, strain_sub_ppm, iso_sub_ppm, reserved}
#pragma once
#include <stdint.h>

#define SYNTH_SEG 10000u

#if defined(__HIPCC__)
#define SYNTH_FN __host__ __device__ inline
#else
#define SYNTH_FN static inline
#endif

SYNTH_FN uint64_t synth_splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}
SYNTH_FN uint64_t synth_h(uint64_t seed, uint64_t x) { return synth_splitmix64(seed ^ (x * 0xD1342543DE82EF95ULL)); }

// 2-bit base code (A=0 C=1 G=2 T=3) at genome-linear position pos
SYNTH_FN uint32_t synth_base(uint64_t species, uint64_t strain, uint64_t isolate, uint32_t acc_pct, uint32_t strain_ppm,
                             uint32_t iso_ppm, uint64_t pos)
{
    uint32_t b = (uint32_t)(synth_h(species, pos) & 3u);
    if (synth_h(strain ^ 0xACCE55ULL, pos / SYNTH_SEG) % 100u < acc_pct) b = (uint32_t)(synth_h(strain ^ 0xACC0BA5EULL, pos) & 3u);
    uint64_t hs = synth_h(strain ^ 0x5B57ULL, pos);
    if (hs % 1000000u < strain_ppm) b = (b + 1u + (uint32_t)((hs >> 32) % 3u)) & 3u;
    uint64_t hi = synth_h(isolate ^ 0x150ULL, pos);
    if (hi % 1000000u < iso_ppm) b = (b + 1u + (uint32_t)((hi >> 32) % 3u)) & 3u;
    return b;
}
