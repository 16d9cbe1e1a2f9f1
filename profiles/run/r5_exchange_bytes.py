"""round 5: bytes every rank SENDS in the two N > 1 exchanges, computed from the generator's recipe (CPU only; no GPU needed):
  replicate   -- raw sketches all-gathered: a rank's seeds (8 B) and markers (8 B) go to world - 1 peers;
  components  -- markers all-gathered (8 B per marker to world - 1 peers), candidate pairs all-gathered (8 B per pair), every genome's seeds
                 (8 B each) ONCE to the rank that owns its connected component (multigpu.component_owners), nothing if that is the sender.
Seeds per genome = bases / 125.4, markers = bases / 1000 (measured densities); candidate pairs = the within-species pairs (what the
screen at 80 % passes on these sets)."""
import json, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from skder_amd import multigpu, synth

def case(name, n, **kw):
    rec = synth.make_recipe(n, **kw)
    L = np.array([rec.total_len(g) for g in range(n)], np.float64)
    n_seeds = (L / 125.4).astype(np.int64)
    n_mark = (L / 1000.0).astype(np.int64)
    per = n // (int(rec.species.max()) + 1)
    ref, query = [], []
    for s0 in range(0, n, per):
        i, j = np.triu_indices(per, 1)
        ref.append(i + s0); query.append(j + s0)
    ref, query = np.concatenate(ref), np.concatenate(query)
    out = {"genomes": n, "bases": int(L.sum()), "seeds": int(n_seeds.sum()), "markers": int(n_mark.sum()), "candidate_pairs": int(len(ref)), "worlds": {}}
    for world in (2, 4, 8):
        blocks = multigpu.partition(n, world)
        owner = multigpu.component_owners(n, ref, query, n_seeds, world)
        rep, comp, held, load = [], [], [], []
        for r, b in enumerate(blocks):
            sl = slice(b.start, b.stop)
            rep.append(int(8 * (n_seeds[sl].sum() + n_mark[sl].sum()) * (world - 1)))
            away = (owner[sl] >= 0) & (owner[sl] != r)
            pairs_mine = int(((ref % world) == r).sum())            # rows r, r + world, ...: an upper bound of a rank's share is fine here
            comp.append(int(8 * n_mark[sl].sum() * (world - 1) + 8 * n_seeds[sl][away].sum() + 8 * pairs_mine * (world - 1)))
            held.append(int(n_seeds[owner == r].sum()))
            sel = owner[ref] == r
            load.append(float((n_seeds[ref[sel]] + n_seeds[query[sel]]).sum()))
        out["worlds"][str(world)] = {"replicate_bytes_sent_per_rank_max": max(rep), "components_bytes_sent_per_rank_max": max(comp),
                                     "components_seeds_held_per_rank_max": max(held), "replicate_seeds_held_per_rank": int(n_seeds.sum()),
                                     "chain_load_imbalance_max_over_mean": max(load) / (sum(load) / world)}
    return name, out

res = dict([case("headline_5000x3Mb", 5000, genome_len=3_000_000), case("config5_50000x1-8Mb", 50000, len_range=(1_000_000, 8_000_000))])
json.dump(res, open(os.path.join("profiles", "round5_exchange_bytes.json"), "w"), indent=1)
for k, v in res.items():
    print(k, {w: (round(x["replicate_bytes_sent_per_rank_max"] / 1e9, 3), round(x["components_bytes_sent_per_rank_max"] / 1e9, 3), round(x["chain_load_imbalance_max_over_mean"], 3)) for w, x in v["worlds"].items()})
