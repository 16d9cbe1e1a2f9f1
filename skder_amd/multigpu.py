"""Row-block sharding of the pair matrix over the GPUs of one node.

One process per GPU.  Each rank sketches a contiguous block of genomes, the raw sketches
(position-ordered seeds: k-mer + position, 8 B/seed; sorted markers: 8 B/marker; record tables) are all-gathered
with torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).
`triangle_sharded` then shares the work out so that nothing big is done twice:
  * rank r builds the k-mer bucket index only for the genomes it OWNS (g mod world == r) -- every genome's chunk
    tables, which are cheap, everywhere;
  * the marker screen is dealt out by rows (i = r, r + world, ...: row i has N-1-i entries);
  * every candidate pair goes to the rank that owns the genome it PROBES, which chains it;
  * edge records are gathered on rank 0.
Collectives: the sketch all-gather (the only large one: one padded collective of everything a rank has, behind a 40-byte header),
the repetitive-cut-off table (4 B per genome), the candidate pairs (8 B per pair, all_to_all_single: each pair to its owner only)
and the edge gather.

`triangle_by_components` (SKDER_AMD_EXCHANGE=components in bench.py) is the exchange that does NOT replicate the seeds: only the
MARKERS are all-gathered (8 B per 1000 bases), every rank screens its rows of the triangle, the candidate pairs are all-gathered
(8 B each), every connected component of the candidate-pair graph -- a species -- is given to ONE rank (heaviest first to the least
loaded rank, weight = the seeds its pairs read), and each genome's seeds travel ONCE, to the rank that owns its component, by one
all_to_all_single per array.  A rank then holds, indexes and chains only its components' genomes: 1/world of the seeds instead of all
of them, and no genome of another rank is needed while chaining.  A component heavier than a rank's fair share (one species holding most
of the genomes) is not an atom: `component_plan` shares it among several ranks, its pairs dealt by probed genome, and every sharing rank
receives the seeds of the genomes its pairs touch (round 6)."""
from typing import Dict, List

import numpy as np
import torch
import torch.distributed as dist


def partition(n: int, world: int) -> List[range]:
    """contiguous, balanced blocks of genome indices"""
    cuts = [(n * r) // world for r in range(world + 1)]
    return [range(cuts[r], cuts[r + 1]) for r in range(world)]


def exchange_raw(raw: Dict, group=None, staging: str = None, parts: bool = False) -> Dict:
    """raw: dict with torch tensors seed_kmer/seed_gpos (int32 views of u32) and markers
    (int64 view of u64) of THIS rank's genomes, and numpy metadata seed_off, marker_off, genome_len,
    genome_nrec, rec_goff.  Returns the same dict for the concatenation of all ranks' genomes.
    TWO collectives: a 40-byte header per rank (the section sizes), then ONE padded all-gather of everything a rank has --
    its per-genome tables, seed k-mers, seed positions and markers packed into one int64 buffer on the device (RCCL moves
    it over xGMI; only the few KB of tables come back to the host).
    staging="cpu": the buffer travels through host memory (gloo backend).
    parts=True: the result also carries "parts", one raw dict per rank whose tensors are VIEWS of the gathered buffer (for
    sketches_from_raw: no concatenation copy); the concatenated tensors are then left out."""
    world = dist.get_world_size(group)
    nccl = dist.get_backend(group) == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if nccl else torch.device("cpu")
    ng = int(raw["n_genomes"])
    tabs = [np.asarray(raw["seed_off"], np.uint64), np.asarray(raw["marker_off"], np.uint64),
            np.asarray(raw["genome_len"], np.uint64), np.asarray(raw["genome_nrec"], np.uint64),
            np.asarray(raw["rec_goff"], np.uint64)]
    blob = np.concatenate(tabs).astype(np.int64)
    ns, nm = int(raw["seed_kmer"].numel()), int(raw["markers"].numel())
    hdr = torch.tensor([ng, ns, nm, len(tabs[4]), len(blob)], dtype=torch.int64, device=dev)
    hdrs = torch.empty(world * 5, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(hdrs, hdr, group=group)
    H = hdrs.cpu().numpy().reshape(world, 5)
    # sections of a rank's buffer, in int64 words: tables | seed k-mers (two per word) | seed positions | markers
    words = lambda h: (int(h[4]), (int(h[1]) + 1) // 2, (int(h[1]) + 1) // 2, int(h[2]))
    mx = max(max(sum(words(H[r])) for r in range(world)), 1)
    wdev = raw["seed_kmer"].device if (nccl or staging != "cpu") else torch.device("cpu")
    mine = torch.empty(mx, dtype=torch.int64, device=wdev)
    w = words(H[dist.get_rank(group)])
    o = 0
    mine[o:o + w[0]] = torch.from_numpy(blob).to(wdev); o += w[0]
    for key in ("seed_kmer", "seed_gpos"):
        if ns & 1:
            mine[o + w[1] - 1] = 0                # the odd half word
        mine[o:o + w[1]].view(torch.int32)[:ns] = raw[key].to(wdev); o += w[1]
    mine[o:o + w[3]] = raw["markers"].to(wdev); o += w[3]
    mine[o:].zero_()                               # padding up to the longest rank's buffer
    if not nccl:
        mine = mine.cpu()
    allb = torch.empty(world * mx, dtype=torch.int64, device=mine.device)
    dist.all_gather_into_tensor(allb, mine, group=group)
    metas, parts_t = [], {"seed_kmer": [], "seed_gpos": [], "markers": []}
    for r in range(world):
        g, nr = int(H[r, 0]), int(H[r, 3])
        wr = words(H[r])
        base = r * mx
        b = allb[base:base + wr[0]].cpu().numpy().view(np.uint64)
        o = 0
        m = dict(n_genomes=g, n_seeds=int(H[r, 1]), n_markers=int(H[r, 2]))
        for key, ln, dt in (("seed_off", g + 1, np.uint64), ("marker_off", g + 1, np.uint64), ("genome_len", g, np.uint64),
                            ("genome_nrec", g, np.uint32), ("rec_goff", nr, np.uint32)):
            m[key] = b[o:o + ln].astype(dt)
            o += ln
        metas.append(m)
        o = base + wr[0]
        parts_t["seed_kmer"].append(allb[o:o + wr[1]].view(torch.int32)[:m["n_seeds"]]); o += wr[1]
        parts_t["seed_gpos"].append(allb[o:o + wr[2]].view(torch.int32)[:m["n_seeds"]]); o += wr[2]
        parts_t["markers"].append(allb[o:o + wr[3]])
    out = dict(n_genomes=sum(m["n_genomes"] for m in metas))
    # the record index of a seed follows from its position and the record table: it is not exchanged
    tens = parts_t
    if parts:
        out["parts"] = [dict(n_genomes=m["n_genomes"], seed_kmer=tens["seed_kmer"][r].to(raw["seed_kmer"].device),
                             seed_gpos=tens["seed_gpos"][r].to(raw["seed_kmer"].device), markers=tens["markers"][r].to(raw["seed_kmer"].device),
                             seed_off=m["seed_off"], marker_off=m["marker_off"], genome_len=m["genome_len"], genome_nrec=m["genome_nrec"],
                             rec_goff=m["rec_goff"]) for r, m in enumerate(metas)]
    else:
        for key in tens:
            out[key] = torch.cat(tens[key]).to(raw[key].device)
    so, mo = [np.zeros(1, np.uint64)], [np.zeros(1, np.uint64)]
    sbase = mbase = np.uint64(0)
    for m in metas:
        so.append(m["seed_off"][1:] - m["seed_off"][0] + sbase)
        mo.append(m["marker_off"][1:] - m["marker_off"][0] + mbase)
        sbase = sbase + np.uint64(m["n_seeds"])
        mbase = mbase + np.uint64(m["n_markers"])
    out["seed_off"] = np.concatenate(so)
    out["marker_off"] = np.concatenate(mo)
    out["genome_len"] = np.concatenate([m["genome_len"] for m in metas])
    out["genome_nrec"] = np.concatenate([m["genome_nrec"] for m in metas])
    out["rec_goff"] = np.concatenate([m["rec_goff"] for m in metas])
    return out


_PINNED = {}


def _pinned(nbytes: int) -> torch.Tensor:
    """a grow-only page-locked host buffer (device -> host copies into pageable memory run at a fifth of the speed)"""
    t = _PINNED.get("buf")
    if t is None or t.numel() < nbytes:
        t = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, pin_memory=True)
        _PINNED["buf"] = t
    return t[:nbytes]


def gather_edges(edges: np.ndarray, group=None, copy: bool = True) -> np.ndarray:
    """edge records of all ranks on rank 0 (others get an empty array).  The records travel as raw bytes:
    one padded gather to rank 0 (device tensors under RCCL, host tensors under gloo), rank order kept.
    copy=False (RCCL): the result is a view of this module's page-locked buffer, valid until the next call."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    nccl = dist.get_backend(group) == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if nccl else torch.device("cpu")
    item = edges.dtype.itemsize
    n = torch.tensor([len(edges)], dtype=torch.int64, device=dev)
    sizes = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(sizes, n, group=group)
    counts = [int(x) for x in sizes.cpu().tolist()]
    mx = max(max(counts), 1)
    buf = torch.empty(mx * item, dtype=torch.uint8, device=dev)
    if len(edges):
        raw = np.ascontiguousarray(edges).view(np.uint8).reshape(-1)
        buf[: raw.size].copy_(torch.from_numpy(raw))
    out = [torch.empty(mx * item, dtype=torch.uint8, device=dev) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, out, dst=0, group=group)
    if rank != 0:
        return edges[:0].copy()
    total = sum(counts) * item
    if total == 0:
        return np.empty(0, dtype=edges.dtype)
    res = np.empty(sum(counts), dtype=edges.dtype) if (copy or not nccl) else None
    if nccl:
        host = _pinned(total)
        o = 0
        for r in range(world):
            k = counts[r] * item
            host[o:o + k].copy_(out[r][:k], non_blocking=True)
            o += k
        torch.cuda.synchronize()
        if not copy:
            return host.numpy().view(edges.dtype)
        res.view(np.uint8).reshape(-1)[:] = host.numpy()
    else:
        o = 0
        flat = res.view(np.uint8).reshape(-1)
        for r in range(world):
            k = counts[r] * item
            flat[o:o + k] = out[r][:k].numpy()
            o += k
    return res


def raw_from_sketches(sk, copy: bool = False) -> Dict:
    """a Sketches object's raw arrays as torch tensors on the current device: views of the set's own memory (copy=False: no copy; valid
    ONLY while the set lives and is not appended to -- the dict keeps a reference to the set under "owner" so that it cannot be collected
    while the views are in use, but closing it by hand is the caller's mistake to avoid), or copies with copy=True"""
    from .engine import device_view, download_tensor
    v = sk.view()
    torch.cuda.synchronize()          # the set's arrays were written on the library's stream, torch reads them on its own
    get = (lambda p, n, dt: download_tensor(p, n, dt, sk.ctx)) if copy else (lambda p, n, dt: device_view(p, n, dt, sk.ctx.device))
    return dict(n_genomes=v["n_genomes"],
                seed_kmer=get(v["d_seed_kmer"], v["n_seeds"], torch.int32),
                seed_gpos=get(v["d_seed_gpos"], v["n_seeds"], torch.int32),
                markers=get(v["d_markers"], v["n_markers"], torch.int64),
                seed_off=v["seed_off"], marker_off=v["marker_off"], genome_len=v["genome_len"],
                genome_nrec=v["genome_nrec"], rec_goff=v["rec_goff"], owner=None if copy else sk)


def sketches_from_raw(ctx, raw: Dict):
    """a new Sketches object holding the genomes described by `raw` (tensors on ctx's device).  A dict from exchange_raw(parts=True) is
    appended rank by rank straight out of the gathered buffer (one copy into the set, none in between)."""
    from .engine import Sketches
    s = Sketches(ctx)
    torch.cuda.synchronize()
    parts = raw.get("parts") or [raw]
    if len(parts) > 1:
        s.reserve(sum(int(p["seed_kmer"].numel()) for p in parts), sum(int(p["markers"].numel()) for p in parts))
    for p in parts:
        if int(p["n_genomes"]) == 0:
            continue
        s.append_raw(p["n_genomes"], p["seed_kmer"].data_ptr(), p["seed_gpos"].data_ptr(), None,
                     p["markers"].data_ptr(), p["seed_off"], p["marker_off"], p["genome_len"], p["genome_nrec"], p["rec_goff"])
    return s


def route_pairs(ref: np.ndarray, query: np.ndarray, probed: np.ndarray, world: int, rank: int, group=None):
    """every candidate pair to the rank that owns the genome it probes (owner = genome mod world): the pairs are ordered by
    destination on the device, the ranks exchange their counts, then ONE all_to_all_single moves each pair (8 bytes) to its owner --
    nobody receives a pair it does not chain.  Received pairs come grouped by sender, each sender's in its own order."""
    nccl = dist.get_backend(group) == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if nccl else torch.device("cpu")
    packed = torch.from_numpy((ref.astype(np.int64) << 32) | query.astype(np.int64)).to(dev)
    dest = torch.from_numpy((probed.astype(np.int64) % world)).to(dev)
    order = torch.argsort(dest, stable=True)
    send = packed[order].contiguous()
    counts = torch.bincount(dest, minlength=world).to(torch.int64)
    incoming = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_to_all_single(incoming, counts, group=group)
    n_in, n_out = [int(x) for x in incoming.cpu().tolist()], [int(x) for x in counts.cpu().tolist()]
    recv = torch.empty(sum(n_in), dtype=torch.int64, device=dev)
    dist.all_to_all_single(recv, send, output_split_sizes=n_in, input_split_sizes=n_out, group=group)
    got = recv.cpu().numpy()
    return (got >> 32).astype(np.uint32), (got & 0xFFFFFFFF).astype(np.uint32)


def triangle_sharded(sk, rank: int, world: int, screen_pct: float, group=None, copy: bool = True) -> np.ndarray:
    """this rank's share of the all-pairs table of `sk` (a set holding ALL genomes, not indexed yet): see the module text"""
    import os
    import sys
    import time
    dbg = os.environ.get("SKDER_AMD_DEBUG") is not None
    t = [time.perf_counter()]

    def lap():
        t.append(time.perf_counter())

    n = sk.view()["n_genomes"]
    owned = (np.arange(n) % world == rank).astype(np.uint8)
    sk.index_part(owned)          # enqueued on the second queue: the marker screen below runs beside it
    lap()
    ref, query = sk.screen_rows(rank, world, screen_pct)
    lap()
    nccl = dist.get_backend(group) == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if nccl else torch.device("cpu")
    # repetitive-k-mer cut-offs: every rank knows its own genomes' (others: 0xFFFFFFFF), the minimum is the table
    rc = torch.from_numpy(sk.rep_cuts(n).astype(np.int64)).to(dev)          # (waits for the index build)
    dist.all_reduce(rc, op=dist.ReduceOp.MIN, group=group)
    sk.set_rep_cuts(rc.cpu().numpy().astype(np.uint32))
    lap()
    probed = sk.pairs_probed(ref, query)
    ref, query = route_pairs(ref, query, probed, world, rank, group)
    lap()
    edges = sk.chain_pairs(ref, query, copy=copy)
    lap()
    names = ("index_enqueue", "screen", "index_wait_rep_cuts", "route", "chain")
    triangle_sharded.last_stage_ms = {k: 1e3 * (t[i + 1] - t[i]) for i, k in enumerate(names)}      # (bench.py --gpus N reports them per rank)
    if dbg and rank == 0:
        print("[skder_amd] triangle_sharded: " + ", ".join("%s %.2f ms" % kv for kv in triangle_sharded.last_stage_ms.items()), file=sys.stderr)
    return edges


# ---------------------------------------------------------------------------------------------------------------------------------
# ownership by connected component: markers to everyone, seeds only to the rank that chains them


def component_labels(n: int, ref: np.ndarray, query: np.ndarray, device=None) -> np.ndarray:
    """label[g] = smallest genome index of g's connected component in the graph whose edges are the candidate pairs.
    device = a GPU: label propagation with pointer jumping there (every node takes the smallest label among itself and its neighbours,
    then its label's label, until nothing moves -- a species is a near-clique: two or three sweeps of two scatter-min operations over
    the pairs, microseconds each); otherwise scipy's connected_components when it is there, else the same propagation in numpy"""
    if len(ref) == 0:
        return np.arange(n, dtype=np.int64)
    if device is not None and torch.device(device).type == "cuda":
        lab = torch.arange(n, dtype=torch.int64, device=device)
        r = torch.from_numpy(np.ascontiguousarray(ref, np.int64)).to(device)
        q = torch.from_numpy(np.ascontiguousarray(query, np.int64)).to(device)
        while True:
            low = torch.minimum(lab[r], lab[q])
            new = lab.clone()
            new.scatter_reduce_(0, r, low, "amin")
            new.scatter_reduce_(0, q, low, "amin")
            new = new[new]
            if torch.equal(new, lab):
                return lab.cpu().numpy()
            lab = new
    lab = np.arange(n, dtype=np.int64)
    ref, query = ref.astype(np.int64), query.astype(np.int64)
    try:
        from scipy.sparse import coo_matrix
        from scipy.sparse.csgraph import connected_components
        _, c = connected_components(coo_matrix((np.ones(len(ref), np.int8), (ref, query)), shape=(n, n)), directed=False)
        first = np.full(int(c.max()) + 1, n, np.int64)
        np.minimum.at(first, c, lab)                       # smallest member of every component
        return first[c]
    except ImportError:
        pass
    while True:
        low = np.minimum(lab[ref], lab[query])
        new = lab.copy()
        np.minimum.at(new, ref, low)
        np.minimum.at(new, query, low)
        new = new[new]
        if np.array_equal(new, lab):
            return lab
        lab = new


def probe_rank(genome_len: np.ndarray, genome_nrec: np.ndarray, n_seeds: np.ndarray, n_markers: np.ndarray) -> np.ndarray:
    """The library's orientation rule (chain.hip chunk_the_query: the less contiguous genome of a pair is cut into chunks, the other one
    is PROBED) as one number per genome: genomes ordered by (total length x mean record length, seed count, marker count) -- the same
    IEEE double operations as the C++ -- with equal keys sharing a rank.  Of a pair (ref, query) the query is chunked iff
    rank[query] <= rank[ref]."""
    t = genome_len.astype(np.float64)
    s = t * (t / np.maximum(genome_nrec, 1).astype(np.float64))
    ns, nm = n_seeds.astype(np.int64), n_markers.astype(np.int64)
    order = np.lexsort((nm, ns, s))
    so, no, mo = s[order], ns[order], nm[order]
    new = np.ones(len(order), bool)
    new[1:] = (so[1:] != so[:-1]) | (no[1:] != no[:-1]) | (mo[1:] != mo[:-1])
    rank = np.empty(len(order), np.int64)
    rank[order] = np.cumsum(new) - 1
    return rank


def probed_genome(ref: np.ndarray, query: np.ndarray, genome_len: np.ndarray, genome_nrec: np.ndarray, n_seeds: np.ndarray,
                  n_markers: np.ndarray) -> np.ndarray:
    """the genome of every pair that is PROBED (the other one is cut into chunks): probe_rank on the per-genome tables every rank holds
    after the marker all-gather"""
    rank = probe_rank(genome_len, genome_nrec, n_seeds, n_markers)
    return np.where(rank[query] <= rank[ref], ref, query).astype(np.int64)


def component_plan(n: int, ref: np.ndarray, query: np.ndarray, n_seeds: np.ndarray, world: int, probed: np.ndarray = None, device=None,
                   rank_of_genome: np.ndarray = None):
    """Who chains which pair, and who must therefore hold which genome's seeds: (pair_rank[k], holds[world, n] bool).

    Components of the candidate-pair graph by descending weight (sum over their pairs of the two genomes' seed counts: what chaining
    them reads), ties by label.  A component no heavier than a rank's fair share (total / world) is an atom: all its pairs to the least
    loaded rank (ties: the lowest).  A HEAVIER one -- one species holding most of the genomes, the shape of the reference's published
    low_mem_greedy workload (README.md:27) -- goes back to the replicate rule INSIDE the component: it is shared by the
    ceil(weight / fair) least loaded ranks, its pairs dealt out by probed genome (probed mod the number of sharing ranks, so that the
    pairs probing one genome stay together, as the join wants them), and every sharing rank holds the genomes its pairs touch.
    The probed genome of a pair comes from `probed` (one entry per pair) or, computed only for the pairs of components that are split,
    from `rank_of_genome` (probe_rank); with neither, nothing is split.  A pure function of its arguments: every rank computes the
    same plan.  (Written for the per-step cost: a handful of passes over the pairs, no sort -- 247,500 pairs in ~3 ms.)"""
    pair_rank = np.full(len(ref), -1, np.int64)
    holds = np.zeros((world, n), bool)
    if len(ref) == 0:
        return pair_rank, holds
    ref, query = np.asarray(ref, np.int64), np.asarray(query, np.int64)
    lab = component_labels(n, ref, query, device)
    plab = lab[ref]                                                         # (a pair's two genomes carry one label)
    # a component's weight = sum over its pairs of both genomes' seeds = sum over its genomes of seeds x pairs the genome is in
    deg = np.bincount(ref, minlength=n) + np.bincount(query, minlength=n)
    w_label = np.bincount(lab, weights=n_seeds.astype(np.float64) * deg, minlength=n)
    labels = np.flatnonzero(np.bincount(lab, minlength=n) >= 2)            # a component with two genomes or more has pairs
    weight = w_label[labels]
    fair = float(weight.sum()) / world
    can_split = world > 1 and (probed is not None or rank_of_genome is not None)
    load = np.zeros(world)
    owner_of_label = np.full(n, -1, np.int64)                              # atoms: every pair of the component to this rank
    shared = []
    for k in np.lexsort((labels, -weight)):
        if can_split and weight[k] > fair:
            sel = np.flatnonzero(plab == labels[k])
            share = min(world, int(np.ceil(weight[k] / fair)))
            ranks = np.argsort(load, kind="stable")[:share]               # the least loaded ranks, ties to the lowest
            pr = (np.asarray(probed, np.int64)[sel] if probed is not None else
                  np.where(rank_of_genome[query[sel]] <= rank_of_genome[ref[sel]], ref[sel], query[sel]))
            pair_rank[sel] = ranks[pr % share]
            load += np.bincount(pair_rank[sel], weights=(n_seeds[ref[sel]] + n_seeds[query[sel]]).astype(np.float64), minlength=world)
            shared.append(sel)
        else:
            r = int(np.argmin(load))
            owner_of_label[labels[k]] = r
            load[r] += weight[k]
    atom = owner_of_label[plab]
    pair_rank = np.where(atom >= 0, atom, pair_rank)
    # atoms: a rank holds every genome of its components; split components: the genomes its pairs touch
    g_owner = owner_of_label[lab]
    g = np.flatnonzero(g_owner >= 0)
    holds[g_owner[g], g] = True
    for sel in shared:
        holds[pair_rank[sel], ref[sel]] = True
        holds[pair_rank[sel], query[sel]] = True
    component_plan.last_load = load                                        # chaining load per rank (seeds read), for the caller's statistics
    return pair_rank, holds


def component_owners(n: int, ref: np.ndarray, query: np.ndarray, n_seeds: np.ndarray, world: int, device=None) -> np.ndarray:
    """owner[g] = the rank that chains every pair of g's component when components are atoms (component_plan without a probed table:
    nothing is split), -1 for a genome without a candidate pair (its seeds go nowhere)"""
    pair_rank, holds = component_plan(n, ref, query, n_seeds, world, None, device)
    owner = np.full(n, -1, np.int64)
    r, g = np.nonzero(holds)
    owner[g] = r
    return owner


def _ranges_index(starts: np.ndarray, lens: np.ndarray, device) -> torch.Tensor:
    """the concatenation of arange(starts[i], starts[i] + lens[i]) as an int64 tensor on `device`"""
    lens_t = torch.from_numpy(np.ascontiguousarray(lens, np.int64)).to(device)
    total = int(lens.sum())
    if total == 0:
        return torch.empty(0, dtype=torch.int64, device=device)
    first = torch.from_numpy(np.ascontiguousarray(starts, np.int64)).to(device)
    before = torch.cumsum(lens_t, 0) - lens_t
    return torch.repeat_interleave(first - before, lens_t) + torch.arange(total, dtype=torch.int64, device=device)


def exchange_seeds(raw: Dict, first_genome: int, holds: np.ndarray, n_seeds: np.ndarray, blocks: List[range], group=None, staging: str = None):
    """this rank's genomes are [first_genome, first_genome + raw['n_genomes']); holds[r, g]: rank r chains a pair of genome g and needs
    its seeds (component_plan; a one-dimensional owner table -- owner[g] or -1 -- is accepted as well); n_seeds is the global per-genome
    table.  Every genome's seed k-mers and positions go to each rank that holds it by ONE all_to_all_single per array; the receive
    sizes follow from the global tables, so no counts are exchanged.  Returns (global indices of the genomes received, ascending; their
    k-mers; their positions) -- the arrays hold the genomes back to back in that order."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    holds = np.asarray(holds)
    if holds.ndim == 1:
        holds = holds[None, :] == np.arange(world)[:, None]
    nccl = dist.get_backend(group) == "nccl"
    dev = raw["seed_kmer"].device if (nccl or staging != "cpu") else torch.device("cpu")
    ng = int(raw["n_genomes"])
    mine = np.arange(first_genome, first_genome + ng)
    local_off = (np.asarray(raw["seed_off"], np.uint64) - np.uint64(raw["seed_off"][0])).astype(np.int64) if ng else np.zeros(1, np.int64)
    # by destination, ascending genome inside one; a genome held by several ranks is sent to each
    send_g = [np.flatnonzero(holds[d, mine]) for d in range(world)] if ng else [np.zeros(0, np.int64)] * world
    order = np.concatenate(send_g) if ng else np.zeros(0, np.int64)
    idx = _ranges_index(local_off[order], n_seeds[mine[order]], dev)
    n_out = [int(n_seeds[mine[g]].sum()) for g in send_g]
    to_me = np.where(holds[rank], n_seeds, 0)
    n_in = [int(to_me[blocks[r].start:blocks[r].stop].sum()) for r in range(world)]
    got = []
    for key in ("seed_kmer", "seed_gpos"):
        send = raw[key].to(dev)[idx].contiguous()
        recv = torch.empty(sum(n_in), dtype=send.dtype, device=dev)
        if not nccl:
            send, recv = send.cpu(), recv.cpu()
        dist.all_to_all_single(recv, send, output_split_sizes=n_in, input_split_sizes=n_out, group=group)
        got.append(recv.to(raw[key].device))
    return np.flatnonzero(holds[rank]), got[0], got[1]


def triangle_by_components(ctx, sk, first_genome: int, n_total: int, rank: int, world: int, screen_pct: float, group=None,
                           staging: str = None) -> np.ndarray:
    """this rank's share of the all-pairs table when `sk` holds only the genomes the rank sketched itself (module text).  The edge
    records carry global genome indices; their union over the ranks equals the one-rank table record for record."""
    import time
    from .engine import EDGE_DTYPE, Sketches
    t = [time.perf_counter()]
    lap = lambda: t.append(time.perf_counter())
    raw = raw_from_sketches(sk)
    blocks = partition(n_total, world)
    empty = raw["seed_kmer"][:0]
    mk = exchange_raw(dict(raw, seed_kmer=empty, seed_gpos=empty), group=group, staging=staging, parts=True)["parts"]
    n_seeds = np.concatenate([np.diff(np.asarray(p["seed_off"], np.uint64).astype(np.int64)) for p in mk]).astype(np.int64)
    lap()
    # every genome's markers, no seeds: the set the rows are screened on
    ms = Sketches(ctx)
    torch.cuda.synchronize()
    ms.reserve(0, sum(int(p["markers"].numel()) for p in mk))
    for p in mk:
        if int(p["n_genomes"]):
            ms.append_raw(p["n_genomes"], 0, 0, None, p["markers"].data_ptr(), np.zeros(p["n_genomes"] + 1, np.uint64), p["marker_off"],
                          p["genome_len"], p["genome_nrec"], p["rec_goff"])
    ms.index_part(np.zeros(n_total, np.uint8))
    ref, query = ms.screen_rows(rank, world, screen_pct)
    lap()
    # all candidate pairs to everyone (8 bytes each): the component table must be the same on every rank
    nccl = dist.get_backend(group) == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if nccl else torch.device("cpu")
    packed = torch.from_numpy((ref.astype(np.int64) << 32) | query.astype(np.int64)).to(dev)
    cnt = torch.tensor([len(ref)], dtype=torch.int64, device=dev)
    cnts = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(cnts, cnt, group=group)
    cnts = [int(x) for x in cnts.cpu().tolist()]
    mx = max(max(cnts), 1)
    buf = torch.zeros(mx, dtype=torch.int64, device=dev)
    buf[:len(ref)] = packed
    allp = torch.empty(world * mx, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(allp, buf, group=group)
    allp = allp.cpu().numpy().reshape(world, mx)
    pairs = np.concatenate([allp[r, :cnts[r]] for r in range(world)])
    aref, aquery = (pairs >> 32).astype(np.int64), (pairs & 0xFFFFFFFF).astype(np.int64)
    g_len_all = np.concatenate([p["genome_len"] for p in mk]).astype(np.int64)
    g_nrec_all = np.concatenate([p["genome_nrec"] for p in mk]).astype(np.int64)
    n_markers = np.concatenate([np.diff(np.asarray(p["marker_off"], np.uint64).astype(np.int64)) for p in mk])
    pair_rank, holds = component_plan(n_total, aref, aquery, n_seeds, world, device=dev if nccl else None,
                                      rank_of_genome=probe_rank(g_len_all, g_nrec_all, n_seeds, n_markers))
    lap()
    have, kmer, gpos = exchange_seeds(raw, first_genome, holds, n_seeds, blocks, group=group, staging=staging)
    lap()
    # the set this rank chains on: the genomes its pairs touch, ascending global index, local indices 0 ..
    edges = np.zeros(0, EDGE_DTYPE)
    hb = holds[:, blocks[rank].start:blocks[rank].stop].copy()
    hb[rank] = False                                            # (what a rank keeps for itself does not travel)
    loads = component_plan.last_load if len(aref) else np.zeros(world)
    stats = {"genomes_held": int(len(have)), "seeds_received": int(kmer.numel()),
             "bytes_sent_seeds": int(8 * (hb.sum(axis=0) * n_seeds[blocks[rank].start:blocks[rank].stop]).sum()),
             "bytes_sent_markers": int(8 * int(mk[rank]["markers"].numel()) * (world - 1)), "pairs_all": int(len(pairs)),
             "pairs_mine": int((pair_rank == rank).sum()), "genomes_held_by_several_ranks": int((holds.sum(axis=0) > 1).sum()),
             "chain_load_max_over_mean": float(loads.max() / max(loads.mean(), 1e-30)) if len(aref) else 1.0}
    if len(have):
        m_off = np.concatenate([np.asarray(p["marker_off"], np.uint64).astype(np.int64)[:-1] + b for p, b in
                                zip(mk, np.concatenate([[0], np.cumsum([int(p["markers"].numel()) for p in mk])])[:-1])])
        m_len = np.concatenate([np.diff(np.asarray(p["marker_off"], np.uint64).astype(np.int64)) for p in mk])
        allm = torch.cat([p["markers"] for p in mk])
        markers = allm[_ranges_index(m_off[have], m_len[have], allm.device)].contiguous()
        g_len = np.concatenate([p["genome_len"] for p in mk])[have]
        g_nrec = np.concatenate([p["genome_nrec"] for p in mk])
        all_goff = np.concatenate([p["rec_goff"] for p in mk])
        r_off = np.concatenate([[0], np.cumsum(g_nrec.astype(np.int64) + 1)])
        rl = (g_nrec.astype(np.int64) + 1)[have]
        rec_goff = all_goff[np.repeat(r_off[have] - (np.cumsum(rl) - rl), rl) + np.arange(int(rl.sum()))]
        ls = Sketches(ctx)
        torch.cuda.synchronize()
        ls.append_raw(len(have), kmer.data_ptr(), gpos.data_ptr(), None, markers.data_ptr(),
                      np.concatenate([[0], np.cumsum(n_seeds[have])]).astype(np.uint64), np.concatenate([[0], np.cumsum(m_len[have])]).astype(np.uint64),
                      g_len, g_nrec[have], rec_goff)
        ls.index()
        local = np.full(n_total, -1, np.int64)
        local[have] = np.arange(len(have))
        sel = pair_rank == rank
        e = ls.chain_pairs(local[aref[sel]].astype(np.uint32), local[aquery[sel]].astype(np.uint32), copy=True)
        e["ref"] = have[e["ref"]]
        e["query"] = have[e["query"]]
        edges = e
        ls.close()
    ms.close()
    lap()
    names = ("markers_all_gather", "screen", "pairs_all_gather_components", "seeds_all_to_all", "index_chain")
    triangle_by_components.last_stage_ms = {k: 1e3 * (t[i + 1] - t[i]) for i, k in enumerate(names)}
    triangle_by_components.last_stats = stats
    return edges
