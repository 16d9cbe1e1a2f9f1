"""Synthetic genome sets, bit-identical with skder_amd/csrc/synth.h.

`make_recipe(n)` lays out n genomes as species x strains x isolates with record (contig) lengths;
`bases_numpy(recipe, g)` materialises one genome on the host (tests, CPU baseline);
the device generator is `Engine.synth_fill` (skder_amd_synth_fill)."""
from dataclasses import dataclass
from typing import List

import numpy as np

SEED = 0x5EED5DE22025
SEG = np.uint64(10000)
M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & M64
    x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & M64
    x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & M64
    return x ^ (x >> np.uint64(31))


def synth_h(seed, x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        return splitmix64(np.uint64(seed) ^ (x * np.uint64(0xD1342543DE82EF95)))


def synth_codes(species, strain, isolate, acc_pct, strain_ppm, iso_ppm, pos: np.ndarray) -> np.ndarray:
    """2-bit base codes at genome-linear positions `pos` (uint64 array)."""
    with np.errstate(over="ignore"):
        pos = pos.astype(np.uint64)
        b = (synth_h(species, pos) & np.uint64(3)).astype(np.uint32)
        acc = (synth_h(np.uint64(strain) ^ np.uint64(0xACCE55), pos // SEG) % np.uint64(100)) < np.uint64(acc_pct)
        b = np.where(acc, (synth_h(np.uint64(strain) ^ np.uint64(0xACC0BA5E), pos) & np.uint64(3)).astype(np.uint32), b)
        hs = synth_h(np.uint64(strain) ^ np.uint64(0x5B57), pos)
        sub = (hs % np.uint64(1000000)) < np.uint64(strain_ppm)
        b = np.where(sub, (b + 1 + ((hs >> np.uint64(32)) % np.uint64(3)).astype(np.uint32)) & 3, b)
        hi = synth_h(np.uint64(isolate) ^ np.uint64(0x150), pos)
        sub = (hi % np.uint64(1000000)) < np.uint64(iso_ppm)
        b = np.where(sub, (b + 1 + ((hi >> np.uint64(32)) % np.uint64(3)).astype(np.uint32)) & 3, b)
        return b.astype(np.uint8)


@dataclass
class Recipe:
    n: int
    lineage: np.ndarray      # (n, 3) uint64: species, strain, isolate seeds
    params: np.ndarray       # (n, 4) uint32: acc_pct, strain_ppm, iso_ppm, 0
    rec_lens: List[np.ndarray]   # per genome: record lengths (all >= 500)
    species: np.ndarray      # (n,) species index

    def total_len(self, g: int) -> int:
        return int(self.rec_lens[g].sum())


def make_recipe(n: int, genome_len: int = 3_000_000, n_species: int = None, strains_per_species: int = 10,
                seed: int = SEED, len_range=None) -> Recipe:
    """n genomes: species roots (iid, genome_len +-5 %) x strain ancestors (1-3 % substitutions, 5-10 %
    accessory segments) x isolates (0.01-0.5 % substitutions); log-normal record lengths.
    len_range = (lo, hi): species lengths uniform in [lo, hi] instead (BASELINE.json's mixed 1-8 Mb set)."""
    if n_species is None:
        n_species = max(1, n // 100)
    rng = np.random.RandomState(seed & 0x7FFFFFFF)
    per_species = -(-n // n_species)
    lineage = np.zeros((n, 3), np.uint64)
    params = np.zeros((n, 4), np.uint32)
    species = np.zeros(n, np.int64)
    rec_lens = []
    sp_len = (genome_len * (0.95 + 0.10 * rng.rand(n_species))).astype(np.int64)
    if len_range is not None:
        sp_len = (len_range[0] + (len_range[1] - len_range[0]) * rng.rand(n_species)).astype(np.int64)
    for g in range(n):
        s = g // per_species
        within = g % per_species
        t = within % strains_per_species if per_species >= strains_per_species else within
        species[g] = s
        s_seed = int(splitmix64(np.array([seed ^ (s + 1)], np.uint64))[0])
        t_seed = int(splitmix64(np.array([s_seed ^ (0x1000 + t)], np.uint64))[0])
        u_seed = int(splitmix64(np.array([t_seed ^ (0x2000000 + within)], np.uint64))[0])
        lineage[g] = (s_seed, t_seed, u_seed)
        trng = np.random.RandomState((t_seed ^ (t_seed >> 32)) & 0x7FFFFFFF)
        urng = np.random.RandomState((u_seed ^ (u_seed >> 32)) & 0x7FFFFFFF)
        params[g] = (int(trng.randint(5, 11)), int(trng.randint(10000, 30001)), int(urng.randint(100, 5001)), 0)
        # records: log-normal lengths, median spread so that N50 spans ~10 kb .. whole genome
        L = int(sp_len[s])
        med = float(np.exp(urng.uniform(np.log(2e4), np.log(3e6))))
        lens = []
        left = L
        while left > 0:
            l = int(np.exp(urng.normal(np.log(med), 0.8)))
            l = max(1000, min(l, left))
            if left - l < 1000:
                l = left
            lens.append(l)
            left -= l
        rec_lens.append(np.array(lens, np.uint32))
    return Recipe(n, lineage, params, rec_lens, species)


def bases_numpy(recipe: Recipe, g: int) -> np.ndarray:
    """ASCII bases of genome g, records back to back (what a FASTA reader would hand over)."""
    L = recipe.total_len(g)
    pos = np.arange(L, dtype=np.uint64)
    sp, st, iso = (int(x) for x in recipe.lineage[g])
    acc, sppm, ippm, _ = (int(x) for x in recipe.params[g])
    codes = synth_codes(sp, st, iso, acc, sppm, ippm, pos)
    return np.frombuffer(b"ACGT", np.uint8)[codes]


def write_fasta(recipe: Recipe, g: int, path: str, width: int = 80) -> None:
    bases = bases_numpy(recipe, g)
    off = 0
    with open(path, "wb") as f:
        for r, l in enumerate(recipe.rec_lens[g]):
            f.write((">g%d_rec%d synthetic species %d\n" % (g, r, recipe.species[g])).encode())
            seq = bases[off:off + int(l)]
            off += int(l)
            for i in range(0, len(seq), width):
                f.write(seq[i:i + width].tobytes())
                f.write(b"\n")


# ---- pairs of KNOWN identity (the generator is the ground truth): an independent check of the ANI estimator and of the
# "learned ANI" stand-in outside the 96.4-100 % range the reference's golden tables cover

TRUTH_STRAIN_PPM = (0, 300, 1500, 5000, 10000, 20000, 35000, 50000, 65000, 80000)


def truth_recipe(genome_len: int = 3_000_000, strain_ppm=TRUTH_STRAIN_PPM, iso_ppm: int = 100, seed: int = SEED) -> Recipe:
    """one species, one isolate per strain, NO accessory segments: every position of every genome is homologous to the
    same position of every other one, so the true identity of a pair is simply the fraction of equal bases.  Strain i
    carries strain_ppm[i] substitutions per million (pairs: ~ the sum of the two rates, 99.97 down to ~85 %)."""
    n = len(strain_ppm)
    rec = make_recipe(n, genome_len=genome_len, n_species=1, strains_per_species=n, seed=seed)
    rec.params[:, 0] = 0
    rec.params[:, 1] = np.asarray(strain_ppm, np.uint32)
    rec.params[:, 2] = iso_ppm
    return rec


def true_identity_matrix(recipe: Recipe) -> np.ndarray:
    """fraction of equal bases of every pair of genomes of a truth_recipe (all positions; no indels in the generator)"""
    L = min(recipe.total_len(g) for g in range(recipe.n))
    same = np.zeros((recipe.n, recipe.n), np.int64)
    step = 1 << 20
    for p0 in range(0, L, step):
        pos = np.arange(p0, min(p0 + step, L), dtype=np.uint64)
        codes = [synth_codes(*[int(x) for x in recipe.lineage[g]], *[int(x) for x in recipe.params[g][:3]], pos) for g in range(recipe.n)]
        for a in range(recipe.n):
            for b in range(a + 1, recipe.n):
                same[a, b] += int((codes[a] == codes[b]).sum())
    out = same / float(L)
    return out + out.T + np.eye(recipe.n)


CLUSTERED_RATES_PCT = (0.0, 0.25, 0.5, 1.0, 1.5, 2.5, 4.0)


def clustered_truth_family(genome_len: int = 2_000_000, rates_pct=CLUSTERED_RATES_PCT, shape: float = 0.3, window: int = 1000, seed: int = SEED):
    """Genomes whose substitutions CLUSTER, with the truth known -- the other side of truth_recipe's iid substitutions.  One random
    ancestor; descendant i carries substitutions at a mean rate of rates_pct[i] percent, the rate of every `window`-base stretch
    multiplied by a Gamma(shape, 1/shape) factor (mean 1; shape 0.3: a third of the windows carry 90 % of the changes, the pattern
    recombination leaves in real genomes and the reason a k-mer estimate reads their identity too high).  No indels, one record:
    the true identity of a pair is the fraction of equal bases.  Returns (list of ASCII base arrays, truth matrix).  Host numpy:
    test / bench input, not a device generator."""
    rng = np.random.RandomState(seed & 0x7FFFFFFF)
    anc = rng.randint(0, 4, genome_len).astype(np.uint8)
    nwin = (genome_len + window - 1) // window
    codes = []
    for r in rates_pct:
        mult = rng.gamma(shape, 1.0 / shape, nwin)
        rate = np.minimum(0.6, np.repeat(mult, window)[:genome_len] * (float(r) / 100.0))
        hit = rng.random_sample(genome_len) < rate
        c = anc.copy()
        c[hit] = (c[hit] + rng.randint(1, 4, int(hit.sum())).astype(np.uint8)) & 3
        codes.append(c)
    n = len(codes)
    truth = np.eye(n)
    for a in range(n):
        for b in range(a + 1, n):
            truth[a, b] = truth[b, a] = float((codes[a] == codes[b]).mean())
    lut = np.frombuffer(b"ACGT", np.uint8)
    return [lut[c] for c in codes], truth


def ani_vs_truth(edges, truth: np.ndarray, bins=((99.5, 100.0), (98.0, 99.5), (95.0, 98.0), (90.0, 95.0), (85.0, 90.0))) -> dict:
    """bias / rms (percentage points) of the engine's two ANI figures against the generator's truth, by true-ANI bin:
    `raw` = the chunk-level k-mer estimate (A/N)^(1/15) (skder_edge_t.ani_raw), `model` = what the table prints (after the
    learned-ANI stand-in of include/skder_amd_spec.h).  Pairs the engine did not report count as `missing`."""
    got = {(int(e["ref"]), int(e["query"])): e for e in edges}
    out = {}
    n = truth.shape[0]
    for lo, hi in bins:
        d_raw, d_mod, missing = [], [], 0
        for a in range(n):
            for b in range(a + 1, n):
                t = 100.0 * truth[a, b]
                if not (lo <= t < hi):
                    continue
                e = got.get((a, b))
                if e is None:
                    missing += 1
                    continue
                d_raw.append(100.0 * float(e["ani_raw"]) - t)
                d_mod.append(100.0 * float(e["ani"]) - t)
        r, m = np.array(d_raw), np.array(d_mod)
        out["%g-%g" % (lo, hi)] = {"pairs": len(d_raw), "missing": missing,
                                   "raw_bias": float(r.mean()) if len(r) else None, "raw_rms": float(np.sqrt((r ** 2).mean())) if len(r) else None,
                                   "model_bias": float(m.mean()) if len(m) else None, "model_rms": float(np.sqrt((m ** 2).mean())) if len(m) else None}
    return out


def real_family_plan(anc_rec_lens, per_ancestor: int, seed: int = 4, sub_log10=(-3.7, -1.5), max_events: int = 3):
    """descendants of real assemblies for skder_amd/csrc/descend.hip (engine.Context.descendants): `per_ancestor` descendants of every
    ancestor (anc_rec_lens: the record lengths of each), with the rates of bench._real_descendant -- substitutions log-uniform
    0.02 - 3 %, one short indel per ~12 substitutions, 0 - 3 structural events (inversion / translocation / deletion of 0.5 - 20 kb,
    each inside one record of at least 6 kb and at most a third of it, no two in one record).  Returns an array of engine.DESCENDANT_DTYPE."""
    from .engine import DESCENDANT_DTYPE
    rng = np.random.RandomState(seed)
    n_anc = len(anc_rec_lens)
    out = np.zeros(n_anc * per_ancestor, DESCENDANT_DTYPE)
    k = 0
    for a in range(n_anc):
        lens = np.asarray(anc_rec_lens[a], np.int64)
        big = np.flatnonzero(lens >= 6000)
        for d in range(per_ancestor):
            sub = 10 ** rng.uniform(*sub_log10)
            out[k]["parent"] = a
            out[k]["sub_ppm"] = int(sub * 1e6)
            out[k]["indel_ppm"] = int(sub * 1e6 / 12.0)
            out[k]["seed"] = (np.uint64(seed) << np.uint64(40)) ^ (np.uint64(a) << np.uint64(20)) ^ np.uint64(d) ^ np.uint64(0x9E3779B97F4A7C15)
            ne = min(int(rng.randint(0, max_events + 1)), len(big))
            recs = rng.choice(big, ne, replace=False) if ne else []
            for e, r in enumerate(recs):
                n = int(rng.randint(500, min(20000, int(lens[r]) // 3)))
                s = int(rng.randint(0, lens[r] - n))
                out[k]["ev"][e] = (int(r), int(rng.randint(0, 3)), s, n, int(rng.randint(0, lens[r] - n + 1)))
            out[k]["n_events"] = ne
            k += 1
    return out
