"""N > 1 path on CPU: two processes, backend gloo.  Exercises the genome partition, the padded
variable-length all-gather of raw sketches and the edge gather of skder_amd/multigpu.py -- everything
of the multi-GPU path except the kernels themselves."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _fake_raw(rank, empty=-1):
    rng = np.random.RandomState(100 + rank)
    ng = 0 if rank == empty else 3 + rank
    ns = rng.randint(5, 40, ng).astype(np.int64)
    nm = rng.randint(1, 9, ng).astype(np.int64)
    nrec = rng.randint(1, 4, ng).astype(np.uint32)
    return dict(
        n_genomes=ng,
        seed_kmer=torch.from_numpy(rng.randint(0, 2 ** 30, ns.sum()).astype(np.int32)),
        seed_gpos=torch.from_numpy(rng.randint(0, 2 ** 20, ns.sum()).astype(np.int32)),
        seed_ctg=torch.from_numpy(rng.randint(0, 3, ns.sum()).astype(np.int32)),
        markers=torch.from_numpy(rng.randint(0, 2 ** 40, nm.sum()).astype(np.int64)),
        seed_off=np.concatenate([[0], np.cumsum(ns)]).astype(np.uint64) + np.uint64(7 * rank),   # offsets need not start at 0
        marker_off=np.concatenate([[0], np.cumsum(nm)]).astype(np.uint64),
        genome_len=rng.randint(1000, 9000, ng).astype(np.uint64),
        genome_nrec=nrec,
        rec_goff=np.concatenate([np.zeros(0, np.uint32)] + [np.arange(k + 1, dtype=np.uint32) * 500 for k in nrec]))


def _worker(rank, world, port, q, empty=-1):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from skder_amd import engine, multigpu
    raw = _fake_raw(rank, empty)
    merged = multigpu.exchange_raw(raw)
    # parts=True: one raw dict per rank, views of the gathered buffer -- their concatenation is the merged form, their tables the senders'
    mp = multigpu.exchange_raw(raw, parts=True)
    assert len(mp["parts"]) == world and "seed_kmer" not in mp
    for key in ("seed_kmer", "seed_gpos", "markers"):
        assert torch.equal(torch.cat([p[key] for p in mp["parts"]]), merged[key]), key
    for r, p in enumerate(mp["parts"]):
        want = _fake_raw(r, empty)
        assert p["n_genomes"] == want["n_genomes"]
        for key in ("seed_off", "marker_off", "genome_len", "genome_nrec", "rec_goff"):
            assert np.array_equal(p[key], want[key]), (r, key)
    edges = np.zeros(0 if rank == empty else 2 + rank, engine.EDGE_DTYPE)
    edges["ref"] = rank
    edges["query"] = np.arange(len(edges)) + 10 * rank
    allv = multigpu.gather_edges(edges)
    q.put((rank, {k: (v.numpy() if isinstance(v, torch.Tensor) else v) for k, v in merged.items()}, allv))
    dist.barrier()
    dist.destroy_process_group()


def test_partition_is_balanced_and_complete():
    sys.path.insert(0, ROOT)
    from skder_amd import multigpu
    for n, w in ((5000, 8), (7, 2), (3, 4), (0, 2)):
        parts = multigpu.partition(n, w)
        assert [i for p in parts for i in p] == list(range(n))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_exchange_and_gather_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, merged, allv = q.get(timeout=120)
        res[r] = (merged, allv)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    raws = [_fake_raw(r) for r in range(world)]
    for r in range(world):
        m = res[r][0]
        assert m["n_genomes"] == sum(x["n_genomes"] for x in raws)
        assert "seed_ctg" not in m                         # record indices are derived on the receiving side, not exchanged
        for k in ("seed_kmer", "seed_gpos", "markers"):
            assert np.array_equal(m[k], np.concatenate([x[k].numpy() for x in raws])), k
        # offsets are rebased to the concatenated arrays
        ns0 = int(raws[0]["seed_off"][-1] - raws[0]["seed_off"][0])
        want_so = np.concatenate([raws[0]["seed_off"] - raws[0]["seed_off"][0],
                                  raws[1]["seed_off"][1:] - raws[1]["seed_off"][0] + np.uint64(ns0)])
        assert np.array_equal(m["seed_off"], want_so)
        assert m["marker_off"][-1] == sum(len(x["markers"]) for x in raws)
        assert np.array_equal(m["genome_len"], np.concatenate([x["genome_len"] for x in raws]))
        assert np.array_equal(m["rec_goff"], np.concatenate([x["rec_goff"] for x in raws]))
    assert len(res[0][1]) == 2 + 3 and len(res[1][1]) == 0          # edges land on rank 0 only
    assert sorted(res[0][1]["ref"].tolist()) == [0, 0, 1, 1, 1]


def test_exchange_with_an_empty_rank():
    """three ranks, the middle one without genomes and without edges (fewer genomes than ranks): empty tensors in
    the padded all-gathers and in the edge gather"""
    world, empty = 3, 1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, empty)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, merged, allv = q.get(timeout=120)
        res[r] = (merged, allv)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    raws = [_fake_raw(r, empty) for r in range(world)]
    for r in range(world):
        m = res[r][0]
        assert m["n_genomes"] == sum(x["n_genomes"] for x in raws) == 3 + 5
        for k in ("seed_kmer", "seed_gpos", "markers"):
            assert np.array_equal(m[k], np.concatenate([x[k].numpy() for x in raws])), k
        assert len(m["seed_off"]) == m["n_genomes"] + 1 and m["seed_off"][-1] == len(m["seed_kmer"])
        assert np.array_equal(m["genome_len"], np.concatenate([x["genome_len"] for x in raws]))
        assert np.array_equal(m["rec_goff"], np.concatenate([x["rec_goff"] for x in raws]))
    assert len(res[0][1]) == 2 + 4 and len(res[1][1]) == 0 and len(res[2][1]) == 0
    assert sorted(res[0][1]["ref"].tolist()) == [0, 0, 2, 2, 2, 2]


def _route_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from skder_amd import multigpu
    ref, query, probed = _fake_pairs(rank, world)
    r, qy = multigpu.route_pairs(ref, query, probed, world, rank)
    q.put((rank, r, qy))
    dist.barrier()
    dist.destroy_process_group()


def _fake_pairs(rank, world):
    """candidate pairs of the rows rank, rank + world, ... of a 23-genome triangle; the probed genome is the one with the
    smaller (index * 7) % 23, an arbitrary rule that does not follow the row"""
    n = 23
    ref, query = [], []
    for i in range(rank, n, world):
        for j in range(i + 1, n):
            if (i * j) % 3 != 1:          # "the screen" lets two thirds through
                ref.append(i)
                query.append(j)
    ref, query = np.array(ref, np.uint32), np.array(query, np.uint32)
    probed = np.where((ref * 7) % 23 < (query * 7) % 23, ref, query).astype(np.uint32)
    return ref, query, probed


@pytest.mark.parametrize("world", [2, 3])
def test_pairs_go_to_the_owner_of_the_probed_genome(world):
    """multigpu.route_pairs: every candidate pair ends up on exactly one rank, the owner (genome mod world) of the genome it probes"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000 + world
    procs = [ctx.Process(target=_route_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, a, b = q.get(timeout=120)
        got[r] = (a, b)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = {}
    for r in range(world):
        ref, query, probed = _fake_pairs(r, world)
        for a, b, c in zip(ref, query, probed):
            want[(int(a), int(b))] = int(c) % world
    seen = {}
    for r in range(world):
        for a, b in zip(*got[r]):
            assert (int(a), int(b)) not in seen
            seen[(int(a), int(b))] = r
    assert seen == want and len(want) > 100


# ---- ownership by connected component (multigpu.triangle_by_components): the host logic and the seed exchange


def test_components_and_their_owners():
    """component_labels / component_owners: three cliques of different weight, a chain of pairs and genomes without any pair.  Every
    genome of a component gets ONE owner, genomes without a pair get none, the heaviest components go to different ranks, and the
    table does not depend on the order the pairs are listed in"""
    sys.path.insert(0, ROOT)
    from skder_amd import multigpu
    n = 40
    groups = [list(range(0, 12)), list(range(12, 20)), list(range(20, 25)), [30, 33, 31, 38]]          # the last one a chain 30-33-31-38
    ref, query = [], []
    for g in groups[:3]:
        for i in g:
            for j in g:
                if i < j:
                    ref.append(i); query.append(j)
    for a, b in zip(groups[3][:-1], groups[3][1:]):
        ref.append(min(a, b)); query.append(max(a, b))
    ref, query = np.array(ref), np.array(query)
    n_seeds = np.full(n, 1000, np.int64)
    lab = multigpu.component_labels(n, ref, query)
    for g in groups:
        assert len({int(lab[i]) for i in g}) == 1 and int(lab[g[0]]) == min(g)
    assert all(int(lab[i]) == i for i in (25, 26, 29, 32, 39))
    for world in (1, 2, 3, 8):
        own = multigpu.component_owners(n, ref, query, n_seeds, world)
        for g in groups:
            assert len({int(own[i]) for i in g}) == 1 and 0 <= int(own[g[0]]) < world
        assert all(int(own[i]) == -1 for i in (25, 26, 29, 32, 39))
        if world >= 3:
            assert len({int(own[g[0]]) for g in groups[:3]}) == 3          # heaviest first, each to the least loaded rank
        perm = np.random.RandomState(world).permutation(len(ref))
        assert np.array_equal(own, multigpu.component_owners(n, ref[perm], query[perm], n_seeds, world))


def _seeds_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from skder_amd import multigpu
    raws = [_fake_raw(r) for r in range(world)]
    n = sum(x["n_genomes"] for x in raws)
    cuts = np.concatenate([[0], np.cumsum([x["n_genomes"] for x in raws])])
    blocks = [range(int(cuts[r]), int(cuts[r + 1])) for r in range(world)]
    n_seeds = np.concatenate([np.diff(x["seed_off"].astype(np.int64)) for x in raws])
    owner = (np.arange(n) * 5 + 1) % (world + 1) - 1                      # -1 .. world-1: some genomes go nowhere
    have, kmer, gpos = multigpu.exchange_seeds(raws[rank], int(cuts[rank]), owner, n_seeds, blocks, staging="cpu")
    q.put((rank, have, kmer.numpy(), gpos.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_seeds_go_to_the_owner_of_their_component(world):
    """multigpu.exchange_seeds: every genome's seed arrays arrive on owner[g] and nowhere else, back to back in ascending genome order"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + os.getpid() % 2000 + world
    procs = [ctx.Process(target=_seeds_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, have, kmer, gpos = q.get(timeout=120)
        got[r] = (have, kmer, gpos)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    raws = [_fake_raw(r) for r in range(world)]
    per_genome = []
    for x in raws:
        off = (x["seed_off"] - x["seed_off"][0]).astype(np.int64)
        for g in range(x["n_genomes"]):
            per_genome.append((x["seed_kmer"].numpy()[off[g]:off[g + 1]], x["seed_gpos"].numpy()[off[g]:off[g + 1]]))
    n = len(per_genome)
    owner = (np.arange(n) * 5 + 1) % (world + 1) - 1
    assert (owner == -1).any()
    for r in range(world):
        have, kmer, gpos = got[r]
        assert np.array_equal(have, np.flatnonzero(owner == r)) and len(have) > 0
        assert np.array_equal(kmer, np.concatenate([per_genome[g][0] for g in have]))
        assert np.array_equal(gpos, np.concatenate([per_genome[g][1] for g in have]))


def _one_giant_and_small(seed=3, giant=300, small=(40, 25, 12, 6)):
    """candidate pairs of one giant clique (a species holding most of the genomes) and a few small ones, per-genome tables made up so
    that the probed-genome rule has something to decide on"""
    rng = np.random.RandomState(seed)
    sizes = (giant,) + tuple(small)
    n = sum(sizes) + 5                       # five genomes without any pair at the end
    groups, at = [], 0
    for k in sizes:
        groups.append(np.arange(at, at + k)); at += k
    ref, query = [], []
    for g in groups:
        i, j = np.triu_indices(len(g), 1)
        ref.append(g[i]); query.append(g[j])
    ref, query = np.concatenate(ref), np.concatenate(query)
    glen = rng.randint(2_000_000, 3_000_000, n).astype(np.int64)
    nrec = rng.randint(1, 200, n).astype(np.int64)
    n_seeds = (glen // 125 + rng.randint(-500, 500, n)).astype(np.int64)
    n_mark = (glen // 1000).astype(np.int64)
    return n, groups, ref, query, glen, nrec, n_seeds, n_mark


def test_probed_genome_is_the_librarys_rule():
    """multigpu.probed_genome against a scalar statement of chain.hip's chunk_the_query (the less contiguous genome is chunked)"""
    sys.path.insert(0, ROOT)
    from skder_amd import multigpu
    n, _, ref, query, glen, nrec, n_seeds, n_mark = _one_giant_and_small()
    glen[7] = glen[9]; nrec[7] = nrec[9]                        # ties fall through to the seed counts, then the marker counts
    n_seeds[11] = n_seeds[13]; glen[11] = glen[13]; nrec[11] = nrec[13]; n_mark[11] = n_mark[13] + 1
    got = multigpu.probed_genome(ref, query, glen, nrec, n_seeds, n_mark)
    for k in np.random.RandomState(0).choice(len(ref), 2000, replace=False).tolist() + [int(np.flatnonzero((ref == 7) & (query == 9))[0]),
                                                                                          int(np.flatnonzero((ref == 11) & (query == 13))[0])]:
        r, q = int(ref[k]), int(query[k])
        sq = float(glen[q]) * (float(glen[q]) / max(int(nrec[q]), 1)); sr = float(glen[r]) * (float(glen[r]) / max(int(nrec[r]), 1))
        if sq != sr: cq = sq < sr
        elif n_seeds[q] != n_seeds[r]: cq = n_seeds[q] < n_seeds[r]
        elif n_mark[q] != n_mark[r]: cq = n_mark[q] < n_mark[r]
        else: cq = True
        assert int(got[k]) == (r if cq else q)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_an_oversized_component_is_split_back_into_shares(world):
    """component_plan with one giant component + several small ones (the shape of README.md:27's workload): every pair
    is chained on exactly one rank, the giant's pairs are dealt by probed genome over several ranks with the pairs of one probed genome
    together, the small components stay atoms, a rank holds exactly the genomes its pairs touch, the chaining load is within 1.15 x
    the mean, and the plan does not depend on the order of the pairs"""
    sys.path.insert(0, ROOT)
    from skder_amd import multigpu
    n, groups, ref, query, glen, nrec, n_seeds, n_mark = _one_giant_and_small()
    probed = multigpu.probed_genome(ref, query, glen, nrec, n_seeds, n_mark)
    pair_rank, holds = multigpu.component_plan(n, ref, query, n_seeds, world, probed)
    assert pair_rank.min() >= 0 and pair_rank.max() < world                      # every pair exactly once, somewhere
    giant = np.isin(ref, groups[0])
    assert len(set(pair_rank[giant].tolist())) == world                          # (its weight is > 90 % of everything: all ranks share it)
    for g in np.unique(probed[giant]):                                           # the pairs that probe one genome stay together
        assert len(set(pair_rank[giant & (probed == g)].tolist())) == 1
    for g in groups[1:]:                                                         # small components are atoms
        assert len(set(pair_rank[np.isin(ref, g)].tolist())) == 1
    for r in range(world):
        touched = np.zeros(n, bool)
        touched[ref[pair_rank == r]] = True; touched[query[pair_rank == r]] = True
        assert np.array_equal(holds[r], touched)
    assert not holds[:, -5:].any()                                                # genomes without a pair go nowhere
    w = (n_seeds[ref] + n_seeds[query]).astype(np.float64)
    load = np.bincount(pair_rank, weights=w, minlength=world)
    assert load.max() <= 1.15 * load.mean(), load / load.mean()
    perm = np.random.RandomState(world).permutation(len(ref))
    pr2, h2 = multigpu.component_plan(n, ref[perm], query[perm], n_seeds, world, probed[perm])
    assert np.array_equal(pr2, pair_rank[perm]) and np.array_equal(h2, holds)
    # without a probed table (component_owners' form) nothing is split
    own = multigpu.component_owners(n, ref, query, n_seeds, world)
    assert len(set(own[groups[0]].tolist())) == 1
    # the form triangle_by_components uses: one rank number per genome, the probed genome worked out only for the pairs of split components
    rk = multigpu.probe_rank(glen, nrec, n_seeds, n_mark)
    pr3, h3 = multigpu.component_plan(n, ref, query, n_seeds, world, rank_of_genome=rk)
    assert np.array_equal(pr3, pair_rank) and np.array_equal(h3, holds)
    assert np.allclose(multigpu.component_plan.last_load, load)


def _shared_seeds_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from skder_amd import multigpu
    raws = [_fake_raw(r) for r in range(world)]
    n = sum(x["n_genomes"] for x in raws)
    cuts = np.concatenate([[0], np.cumsum([x["n_genomes"] for x in raws])])
    blocks = [range(int(cuts[r]), int(cuts[r + 1])) for r in range(world)]
    n_seeds = np.concatenate([np.diff(x["seed_off"].astype(np.int64)) for x in raws])
    holds = _fake_holds(n, world)
    have, kmer, gpos = multigpu.exchange_seeds(raws[rank], int(cuts[rank]), holds, n_seeds, blocks, staging="cpu")
    q.put((rank, have, kmer.numpy(), gpos.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def _fake_holds(n, world):
    """genome g is held by rank (g mod world); every third genome by ALL ranks (a shared component's genome); every seventh by none"""
    h = np.zeros((world, n), bool)
    h[np.arange(n) % world, np.arange(n)] = True
    h[:, ::3] = True
    h[:, ::7] = False
    return h


@pytest.mark.parametrize("world", [2, 3])
def test_seeds_of_a_shared_component_reach_every_sharing_rank(world):
    """multigpu.exchange_seeds with a holds matrix: a genome held by several ranks arrives on each of them, one held by none nowhere"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 37500 + os.getpid() % 2000 + world
    procs = [ctx.Process(target=_shared_seeds_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, have, kmer, gpos = q.get(timeout=120)
        got[r] = (have, kmer, gpos)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    raws = [_fake_raw(r) for r in range(world)]
    per_genome = []
    for x in raws:
        off = (x["seed_off"] - x["seed_off"][0]).astype(np.int64)
        for g in range(x["n_genomes"]):
            per_genome.append((x["seed_kmer"].numpy()[off[g]:off[g + 1]], x["seed_gpos"].numpy()[off[g]:off[g + 1]]))
    holds = _fake_holds(len(per_genome), world)
    assert (holds.sum(axis=0) > 1).any() and (holds.sum(axis=0) == 0).any()
    for r in range(world):
        have, kmer, gpos = got[r]
        assert np.array_equal(have, np.flatnonzero(holds[r])) and len(have) > 0
        assert np.array_equal(kmer, np.concatenate([per_genome[g][0] for g in have]))
        assert np.array_equal(gpos, np.concatenate([per_genome[g][1] for g in have]))
