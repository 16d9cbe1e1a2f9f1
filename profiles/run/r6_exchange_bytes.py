"""round 6: bytes every rank SENDS and the seeds it HOLDS in the two N > 1 exchanges, computed from the generator's recipe with the plan the
ranks themselves compute (multigpu.component_plan; CPU only, no GPU needed):
  replicate   -- raw sketches all-gathered: a rank's seeds (8 B) and markers (8 B) go to world - 1 peers;
  components  -- markers all-gathered (8 B per marker to world - 1 peers), candidate pairs all-gathered (8 B per pair), every genome's seeds
                 (8 B each) to each OTHER rank that chains one of its pairs: one rank when its component is an atom, every sharing rank when
                 the component is heavier than a rank's fair share and is split back into shares (round 6).
Seeds per genome = bases / 125.4, markers = bases / 1000 (measured densities); candidate pairs = the within-species pairs (what the
screen at 80 % passes on these sets).  New in round 6: the ONE-SPECIES column (20,000 genomes, every pair a candidate: the shape of the
reference's published low_mem_greedy workload, README.md:27), which the atom-only plan of round 5 gave to a single rank."""
import json, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from skder_amd import multigpu, synth


def case(name, n, one_species=False, **kw):
    rec = synth.make_recipe(n, **kw)
    L = np.array([rec.total_len(g) for g in range(n)], np.float64)
    nrec = np.array([len(rec.rec_lens[g]) for g in range(n)], np.int64)
    n_seeds = (L / 125.4).astype(np.int64)
    n_mark = (L / 1000.0).astype(np.int64)
    per = n if one_species else n // (int(rec.species.max()) + 1)
    ref, query = [], []
    for s0 in range(0, n, per):
        i, j = np.triu_indices(per, 1)
        ref.append((i + s0).astype(np.int64)); query.append((j + s0).astype(np.int64))
    ref, query = np.concatenate(ref), np.concatenate(query)
    probed = multigpu.probed_genome(ref, query, L.astype(np.int64), nrec, n_seeds, n_mark)
    w_pair = (n_seeds[ref] + n_seeds[query]).astype(np.float64)
    out = {"genomes": n, "bases": int(L.sum()), "seeds": int(n_seeds.sum()), "markers": int(n_mark.sum()), "candidate_pairs": int(len(ref)), "worlds": {}}
    for world in (2, 4, 8):
        blocks = multigpu.partition(n, world)
        pair_rank, holds = multigpu.component_plan(n, ref, query, n_seeds, world, probed)
        rep, comp = [], []
        for r, b in enumerate(blocks):
            sl = slice(b.start, b.stop)
            rep.append(int(8 * (n_seeds[sl].sum() + n_mark[sl].sum()) * (world - 1)))
            others = holds[:, sl].copy(); others[r] = False
            pairs_mine = int(((ref % world) == r).sum())            # rows r, r + world, ...: what the rank's screen finds, give or take
            comp.append(int(8 * n_mark[sl].sum() * (world - 1) + 8 * (others.sum(axis=0) * n_seeds[sl]).sum() + 8 * pairs_mine * (world - 1)))
        held = [int(n_seeds[holds[r]].sum()) for r in range(world)]
        load = np.bincount(pair_rank, weights=w_pair, minlength=world)
        out["worlds"][str(world)] = {"replicate_bytes_sent_per_rank_max": max(rep), "components_bytes_sent_per_rank_max": max(comp),
                                     "components_seeds_held_per_rank_max": max(held), "replicate_seeds_held_per_rank": int(n_seeds.sum()),
                                     "genomes_held_by_several_ranks": int((holds.sum(axis=0) > 1).sum()),
                                     "chain_load_imbalance_max_over_mean": float(load.max() / load.mean())}
    return name, out


res = dict([case("headline_5000x3Mb", 5000, genome_len=3_000_000), case("config5_50000x1-8Mb", 50000, len_range=(1_000_000, 8_000_000)),
            case("one_species_20000x2.8Mb", 20000, one_species=True, genome_len=2_800_000, n_species=1)])
json.dump(res, open(os.path.join("profiles", "round6_exchange_bytes.json"), "w"), indent=1)
for k, v in res.items():
    print(k, {w: (round(x["replicate_bytes_sent_per_rank_max"] / 1e9, 3), round(x["components_bytes_sent_per_rank_max"] / 1e9, 3),
                  round(x["components_seeds_held_per_rank_max"] / 1e6, 1), round(x["chain_load_imbalance_max_over_mean"], 3)) for w, x in v["worlds"].items()})
