#!/usr/bin/env python3
"""bench.py -- genome-pairs ANI/sec on the 5,000 x 3 Mb synthetic set (BASELINE.json metric).

One "step" = one pass of the hot path over the whole batch: FracMinHash sketching of every genome
(bases already resident in HBM), index, marker screen, anchors + chaining + ANI/AF of every screened
pair, edge records back on the host.  value = N(N-1)/2 / step time.  With --gpus G > 1 (launched by
torch.distributed.run) each rank sketches N/G genomes, the sketches are all-gathered over RCCL, every rank
indexes the genomes it owns, screens its share of the rows and chains the pairs that probe its genomes
(skder_amd/multigpu.py): total work is fixed, so scaling is "strong".

The JSON line also carries `roofline` (dominant kernel: algorithmic bytes / HIP-event time vs the
8 TB/s HBM peak) and `cpu_baseline` (the CPU oracle timed on a bounded sample on this host)."""
import os as _os
# ROCm maps HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  Under torch.distributed RCCL and torch own
# several streams, and the library's two chaining queues then share ONE hardware queue and stop overlapping (measured:
# the chain stage 45 ms instead of 42); must be set before the HIP runtime initialises.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0
SKETCH_BODY_NS = 93.4      # profiles/round3_sketch_body.json, variant 32781 (SK_BODY_DEFAULT): ns per position and wavefront per SIMD
PMC_TRAFFIC = os.path.join("profiles", "round6_pmc_traffic.json")     # rocprofv3 --pmc passes of THIS build (profiles/collect.sh)
ONE_QUEUE_STEPS = 3


def _write_fasta(tmp, g, host, layout, k):
    """genome k of a batch held in `host` (device layout) as an 80-column FASTA file tmp/g<g>.fasta; -> (path, bytes)"""
    parts = []
    for r in range(int(layout.genome_rec_begin[k]), int(layout.genome_rec_begin[k + 1])):
        o, l = int(layout.rec_off[r]), int(layout.rec_len[r])
        seq = host[o:o + l]
        full = (l // 80) * 80
        body = np.concatenate([seq[:full].reshape(-1, 80), np.full((full // 80, 1), ord("\n"), np.uint8)], axis=1).reshape(-1)
        parts.append((">g%d_rec%d synthetic\n" % (g, r)).encode())
        parts.append(body.tobytes())
        if l > full:
            parts.append(seq[full:].tobytes() + b"\n")
    p = os.path.join(tmp, "g%05d.fasta" % g)
    blob = b"".join(parts)
    with open(p, "wb") as f:
        f.write(blob)
    return p, len(blob)


def write_sample_files(batches, n_sample):
    """FASTA files of the first n_sample genomes of a resident batch [(layout, device tensor)] (the measurement scripts under
    profiles/run write their inputs with this); -> (directory, paths, bytes)"""
    import tempfile
    layout, d = batches[0]
    n_sample = min(n_sample, layout.n_genomes)
    host = d.cpu().numpy()
    tmp = tempfile.mkdtemp(prefix="skder_amd_sample_")
    res = [_write_fasta(tmp, g, host, layout, g) for g in range(n_sample)]
    return tmp, [r[0] for r in res], sum(r[1] for r in res)


def write_workload_sample(engine, ctx, torch, recipe, genomes, chunk=250):
    """FASTA files (80 columns) of the workload's genomes `genomes`, generated on the device again after the timed region (any rank
    can write any genome: at N > 1 rank 0 holds only its own block) and formatted on the host threads; -> (directory, paths, sizes)"""
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    tmp = tempfile.mkdtemp(prefix="skder_amd_sample_")
    paths, sizes = [], []
    genomes = [int(g) for g in genomes]
    with ThreadPoolExecutor(max_workers=max(1, min(16, granted_cpus()))) as ex:
        for c0 in range(0, len(genomes), chunk):
            gs = genomes[c0:c0 + chunk]
            layout = engine.BatchLayout([recipe.rec_lens[g] for g in gs])
            d = torch.empty(layout.total_bytes, dtype=torch.uint8, device="cuda")
            ctx.synth_fill(d.data_ptr(), layout, recipe.lineage[gs], recipe.params[gs])
            torch.cuda.synchronize()
            host = d.cpu().numpy()
            del d
            for p, n in ex.map(lambda kg: _write_fasta(tmp, kg[1], host, layout, kg[0]), list(enumerate(gs))):
                paths.append(p)
                sizes.append(n)
    return tmp, paths, sizes


def gzip_sample_files(paths):
    """the sample files gzip-compressed (level 1; zlib releases the GIL, so a thread pool compresses in parallel): the
    reference's real inputs are .fasta.gz (test_case/skder_gtdb_results/gtdb_ncbi_genomes/)"""
    import gzip
    from concurrent.futures import ThreadPoolExecutor

    def one(p):
        with open(p, "rb") as f:
            blob = gzip.compress(f.read(), compresslevel=1)
        with open(p + ".gz", "wb") as o:
            o.write(blob)
        return p + ".gz", len(blob)
    with ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 1)) as ex:
        res = list(ex.map(one, paths))
    return [r[0] for r in res], sum(r[1] for r in res)


def evict_from_page_cache(paths):
    """the files written back and their pages dropped from the page cache (posix_fadvise DONTNEED: what an ordinary user can do);
    returns the file system the directory lives on (a tmpfs keeps its pages: 'cold' means nothing there)"""
    fstype = "unknown"
    try:
        d, best = os.path.realpath(os.path.dirname(paths[0])), ""
        for line in open("/proc/mounts"):
            f = line.split()
            if len(f) >= 3 and (d == f[1] or d.startswith(f[1].rstrip("/") + "/")) and len(f[1]) >= len(best):
                best, fstype = f[1], f[2]
    except OSError:
        pass
    for q in paths:
        fd = os.open(q, os.O_RDONLY)
        try:
            os.fsync(fd)
            os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)
        finally:
            os.close(fd)
    return fstype


def granted_cpus():
    """CPUs this process may actually use: the cgroup CPU quota if there is one (the MI355X boxes show 256 hardware threads and grant
    16 CPUs' worth of time), else the CPU count"""
    n = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def end_to_end_sample(tmp, paths, nbytes, device):
    """The file-based drop-in on a bounded sample (second clock): skder_amd_triangle runs
    listing -> ingest (read, parse, N50, PCIe copy) -> sketch -> index -> screen -> chain -> TSV on disk."""
    import ctypes as C
    import torch
    from skder_amd import _lib
    listing = os.path.join(tmp, "listing.txt")
    open(listing, "w").write("".join(p + "\n" for p in paths))
    out = os.path.join(tmp, "edges.tsv")
    n50 = os.path.join(tmp, "n50.tsv")
    for f in (out, n50):
        if os.path.exists(f):
            os.remove(f)
    err = C.create_string_buffer(_lib.ERRLEN)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rc = _lib.lib().skder_amd_triangle_n50(listing.encode(), 50.0, 80.0, device, out.encode(), n50.encode(), err, _lib.ERRLEN)
    dt = time.perf_counter() - t0
    if rc != 0:
        raise RuntimeError(err.value.decode())
    rows = sum(1 for _ in open(out)) - 1
    return {"genomes": len(paths), "fasta_bytes": nbytes, "seconds": dt, "rows": rows, "ingest_MB_per_s": nbytes / dt / 1e6}


def pmc_traffic(kernel, path=None):
    """measured HBM traffic of `kernel` per step and of the whole step, from the committed summary of the rocprofv3 --pmc passes
    (FETCH_SIZE / WRITE_SIZE in separate passes, corrected by the calibration of profiles/calib: profiles/summarise.py); only valid for the
    default workload on one GPU.  -> (kernel bytes, where they come from, step bytes); (None, why, None) when the file is not there --
    an auxiliary field never costs the line"""
    src = path or PMC_TRAFFIC
    try:
        allk = json.load(open(os.path.join(ROOT, src)))
        pm = allk[kernel]
        traffic = sum(pm[c].get("corrected_bytes", pm[c]["sum_counter_kb"] * 1024.0) for c in ("FETCH_SIZE", "WRITE_SIZE"))
        source = src + " (static: rocprofv3 --pmc passes, not measured in this run; " + str(allk.get("__source__", "of this build"))[:400] + ")"
        return traffic, source, allk.get("__step__", {}).get("bytes")
    except OSError:
        return None, "missing: %s (collect it with profiles/collect.sh)" % src, None
    except (KeyError, ValueError, TypeError) as ex:
        return None, "unreadable: %s (%r)" % (src, ex), None


def golden_compare(table):
    """a 7-column edge table of the reference's 34 genomes against the skani table the reference's own run holds (tests/golden/G5, two
    decimals): differences in percentage points and -- the distance to "bit-identical edge table" as integers -- how many golden values
    and rows the table reproduces at print precision and how many of the 30 `-tc` representative listings (bin/skder:331-407: 6 ANI x 5
    AF cut-offs) come out identical from it"""
    from skder_amd import selection as S
    gold = os.path.join(ROOT, "tests", "golden")
    want = {}
    with open(os.path.join(gold, "G5_triangle_minaf10_s89.5.tsv")) as f:
        next(f)
        for line in f:
            c = line.rstrip("\n").split("\t")
            want[frozenset((os.path.basename(c[0]), os.path.basename(c[1])))] = (float(c[2]), float(c[3]), float(c[4]), os.path.basename(c[0]))
    d_ani, d_af, seen = [], [], 0
    values_equal = rows_equal = 0
    with open(table) as f:
        next(f)
        for line in f:
            c = line.rstrip("\n").split("\t")
            k = frozenset((os.path.basename(c[0]), os.path.basename(c[1])))
            if k not in want:
                continue
            seen += 1
            g = want[k]
            afr, afq = (float(c[3]), float(c[4])) if os.path.basename(c[0]) == g[3] else (float(c[4]), float(c[3]))
            d_ani.append(float(c[2]) - g[0])
            d_af += [afr - g[1], afq - g[2]]
            eq = [round(100 * float(c[2])) == round(100 * g[0]), round(100 * afr) == round(100 * g[1]), round(100 * afq) == round(100 * g[2])]
            values_equal += sum(eq)
            rows_equal += all(eq)
    D = os.path.join(gold, "downstream")
    tc_same, tc_differ = 0, []
    edges_tc = [(os.path.basename(a), os.path.basename(b), x, y, z) for a, b, x, y, z in S.edges_from_table(table)]
    n50_tc = S.read_n50(os.path.join(D, "skder_gtdb_results__Concatenated_N50.txt"))
    for a in (90.0, 95.0, 97.0, 98.0, 99.0, 99.5):
        for fcut in (10.0, 25.0, 50.0, 75.0, 90.0):
            with open(os.path.join(D, "tc", "skDER_Results_ANI%s_AF%s.txt" % (a, fcut))) as fh:
                wanted = [l.rstrip("\n") for l in fh]
            if S.greedy_from_edges(edges_tc, n50_tc, a, fcut) == wanted:
                tc_same += 1
            else:
                tc_differ.append("ANI%s_AF%s" % (a, fcut))
    d_ani, d_af = np.array(d_ani), np.array(d_af)
    return {"max_abs_dANI": float(np.abs(d_ani).max()), "rms_dANI": float(np.sqrt((d_ani ** 2).mean())),
            "max_abs_dAF": float(np.abs(d_af).max()), "rms_dAF": float(np.sqrt((d_af ** 2).mean())),
            "pairs": seen, "golden_pairs": len(want), "unit": "percentage points",
            "values_equal_at_print_precision": {"equal": values_equal, "of": 3 * len(want), "what": "ANI, AF_ref, AF_query of the golden rows that print the same two decimals"},
            "rows_fully_equal": {"equal": rows_equal, "of": len(want)},
            "tc_listings_identical": {"equal": tc_same, "of": 30, "differ": tc_differ,
                                      "what": "greedy representative listings of the reference's 6 x 5 cut-off sweep from this table vs the reference's 30 files"}}


def golden_parity(device):
    """BASELINE.json's second figure, "max |dANI| vs skani": the drop-in triangle on the 34 genomes of the
    reference's own test run against the skani table that run holds (golden_compare).
    Outside the timed region; skani itself is not available on this box."""
    import ctypes as C
    import tempfile
    from skder_amd import _lib
    gold = os.path.join(ROOT, "tests", "golden")
    names = sorted(os.listdir(os.path.join(gold, "genomes")))
    with tempfile.TemporaryDirectory(prefix="skder_amd_gold_") as tmp:
        listing = os.path.join(tmp, "listing.txt")
        open(listing, "w").write("".join(os.path.join(gold, "genomes", n) + "\n" for n in names))
        out = os.path.join(tmp, "tri.tsv")
        err = C.create_string_buffer(_lib.ERRLEN)
        n50 = os.path.join(tmp, "n50.tsv")
        gz_bytes = sum(os.path.getsize(os.path.join(gold, "genomes", n)) for n in names)
        t0 = time.perf_counter()
        if _lib.lib().skder_amd_triangle_n50(listing.encode(), 10.0, 89.5, device, out.encode(), n50.encode(), err, _lib.ERRLEN) != 0:
            raise RuntimeError(err.value.decode())
        call_s = time.perf_counter() - t0
        res = golden_compare(out)
        # the real engine, if this box has it (oracle/skani_ref.py): the same listing through `skani triangle`, cell by cell
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import skani_ref
        live = "skani unavailable on this host (not on PATH nor in the usual conda locations): the golden table of the reference's own run stands in"
        if skani_ref.find():
            sk_out = os.path.join(tmp, "tri_skani.tsv")
            run = skani_ref.triangle(listing, sk_out, 10.0, 89.5, os.cpu_count() or 1)
            live = skani_ref.compare_tables(out, sk_out)
            live.update({"skani_version": run["version"], "skani_seconds": run["seconds"], "command": run["command"]})
    res.update({"live_skani": live,
                "drop_in_call": {"seconds": call_s, "gz_bytes": gz_bytes, "what": "skder_amd_triangle_n50 on the reference's 34 .fasta.gz files (listing -> N50 table + edge table on disk), one call incl. context creation"},
                "against": "skani table of the reference's own test run (tests/golden/G5: 34 C. granulosum genomes, ANI 96.4-100, two decimals); "
                           "skani's version is unpinned and its learned-ANI model is replaced by a fitted map (DESIGN.md 2)"})
    return res


def _read_fasta_records(path):
    """FASTA (plain or gz) -> (kept record lengths, concatenated kept bases), records >= 500 bp as the product's reader keeps them"""
    import gzip
    op = gzip.open if path.endswith(".gz") else open
    recs, cur = [], None
    with op(path, "rb") as f:
        for line in f:
            if line.startswith(b">"):
                cur = []
                recs.append(cur)
            elif cur is not None:
                cur.append(line.strip())
    kept = [b"".join(r) for r in recs]
    kept = [x for x in kept if len(x) >= 500]
    return np.array([len(x) for x in kept], np.uint32), np.frombuffer(b"".join(kept), np.uint8)


def _indel_descendant(rng, anc):
    """a descendant with substitutions (0.05 - 4 %) and SHORT INDELS, one per ~12 substitutions, geometric lengths
    (mean 2.5): the FUZZ_REAL model of tests/tools (the counter-based device generator has substitutions only)"""
    alpha = np.frombuffer(b"ACGT", np.uint8)
    seq = anc.copy()
    sub = 10 ** rng.uniform(-3.3, -1.4)
    k = rng.binomial(len(seq), sub)
    if k:
        idx = rng.randint(0, len(seq), k)
        seq[idx] = alpha[(np.searchsorted(alpha, seq[idx]) + 1 + rng.randint(0, 3, k)) % 4]
    every = max(150, int(12.0 / sub))
    n_ev = max(1, len(seq) // every)
    at = np.sort(rng.randint(1, len(seq) - 1, n_ev))
    ln = rng.geometric(0.4, n_ev)
    ins = rng.rand(n_ev) < 0.5
    out, pos = [], 0
    for a, n, i in zip(at, ln, ins):
        if a < pos:
            continue
        out.append(seq[pos:a])
        if i:
            out.append(alpha[rng.randint(0, 4, n)])
            pos = a
        else:
            pos = a + n
    out.append(seq[pos:])
    seq = np.concatenate(out)
    nrec = int(10 ** rng.uniform(0, 1.5))
    cuts = np.sort(rng.choice(np.arange(1000, len(seq) - 1000, 1000), size=nrec - 1, replace=False)) if nrec > 1 else np.array([], int)
    lens = np.diff(np.concatenate([[0], cuts, [len(seq)]])).astype(np.uint32)
    return seq, lens


_COMP = np.zeros(256, np.uint8)
for _a, _b in zip(b"ACGTacgtNn", b"TGCAtgcaNn"):
    _COMP[_a] = _b


def _real_descendant(seed, anc, anc_lens):
    """A descendant of a REAL assembly (concatenated kept records `anc`, record lengths `anc_lens`): substitutions
    (log-uniform 0.02 - 3 %), short indels (one per ~12 substitutions, geometric lengths, mean 2.5: the FUZZ_REAL model of
    tests/tools), and 0 - 3 structural events (inversion, translocation or deletion of 0.5 - 20 kb).  Vectorised: one
    np.delete / np.insert pass per genome.  The assembly's record boundaries are carried through the edits."""
    rng = np.random.RandomState(seed)
    alpha = np.frombuffer(b"ACGT", np.uint8)
    seq = anc.copy()
    L = len(seq)
    sub = 10 ** rng.uniform(-3.7, -1.5)
    k = rng.binomial(L, sub)
    if k:
        idx = rng.randint(0, L, k)
        seq[idx] = alpha[(np.searchsorted(alpha, seq[idx] & 0xDF) + 1 + rng.randint(0, 3, k)) % 4]
    bounds = np.concatenate([[0], np.cumsum(anc_lens.astype(np.int64))])
    n_ev = max(1, int(L * sub / 12.0))
    at = np.sort(rng.randint(1, L - 1, n_ev))
    ln = rng.geometric(0.4, n_ev)
    ins = rng.rand(n_ev) < 0.5
    # deletions: a mask of the bases that stay
    d_at, d_ln = at[~ins], ln[~ins]
    diff = np.zeros(L + 1, np.int32)
    np.add.at(diff, d_at, 1)
    np.add.at(diff, np.minimum(d_at + d_ln, L), -1)
    keep = np.cumsum(diff[:L]) == 0
    new_index = np.concatenate([[0], np.cumsum(keep)])          # index of original base i among the kept ones
    seq = seq[keep]
    bounds = new_index[bounds]
    # insertions, positions in the kept sequence
    i_at, i_ln = new_index[at[ins]], ln[ins]
    if len(i_at):
        seq = np.insert(seq, np.repeat(i_at, i_ln), alpha[rng.randint(0, 4, int(i_ln.sum()))])
        shift = np.concatenate([[0], np.cumsum(i_ln)])
        bounds = bounds + shift[np.searchsorted(i_at, bounds, side="left")]
    for _ in range(rng.randint(0, 4)):
        if len(seq) < 100000:
            break
        n, ev = rng.randint(500, 20000), rng.randint(0, 3)
        a = rng.randint(0, len(seq) - n - 1)
        seg = seq[a:a + n]
        if ev == 0:
            seq[a:a + n] = _COMP[seg[::-1]]
        elif ev == 1:
            rest = np.concatenate([seq[:a], seq[a + n:]])
            b = rng.randint(0, len(rest))
            seq = np.concatenate([rest[:b], seg, rest[b:]])
        else:
            seq = np.concatenate([seq[:a], seq[a + n:]])
            bounds = np.where(bounds > a + n, bounds - n, np.minimum(bounds, a))
    bounds[-1] = len(seq)
    lens = np.diff(np.maximum.accumulate(np.minimum(bounds, len(seq))))
    keep_l = []
    for l in lens:                                  # records below 500 bases join their neighbour
        if keep_l and (l < 500 or keep_l[-1] < 500):
            keep_l[-1] += l
        else:
            keep_l.append(l)
    return seq, np.array(keep_l, np.uint32)


def _triangle_stats(engine, ctx, torch, rec_lens_list, bases_list, screen, steps=2):
    """sketch + triangle of host genomes through the device API; per-step times of the chaining stage and path counters"""
    layout = engine.BatchLayout(rec_lens_list)
    d = torch.from_numpy(layout.pack_host(bases_list)).cuda()
    sk = engine.Sketches(ctx)
    sk.sketch_batch(d.data_ptr(), layout)
    best = None
    # the stage times below are sums of per-kernel event brackets: keep the batches on one queue (on two, the brackets of
    # overlapping batches include each other's share of the chip)
    queues_env = os.environ.get("SKDER_AMD_QUEUES")
    os.environ["SKDER_AMD_QUEUES"] = "1"
    for _ in range(steps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        edges = sk.triangle_rows(0, 1, screen, copy=False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        t, c = ctx.timing(), ctx.counters()
        cur = {"triangle_ms": dt * 1e3, "chained_pairs": int(t[6]), "edges": int(len(edges)), "chunks": int(c[0]), "slow_path_chunks": int(c[1]),
               "join_ms": float(c[2]) / 1000.0, "run_extract_ms": ctx.runs_ms(), "chain_fast_ms": float(t[3]), "chain_slow_ms": float(t[4]),
               "finalize_ms": float(t[5])}
        if best is None or cur["triangle_ms"] < best["triangle_ms"]:
            best = cur
    sk.close()
    del d
    if queues_env is None:
        del os.environ["SKDER_AMD_QUEUES"]
    else:
        os.environ["SKDER_AMD_QUEUES"] = queues_env
    n = len(rec_lens_list)
    stage = best["join_ms"] + best["run_extract_ms"] + best["chain_fast_ms"] + best["chain_slow_ms"] + best["finalize_ms"]
    best.update({"genomes": n, "pairs": n * (n - 1) // 2, "slow_path_fraction": best["slow_path_chunks"] / max(best["chunks"], 1),
                 "chain_stage_ms": stage, "us_per_chained_pair": 1e3 * stage / max(best["chained_pairs"], 1),
                 "pairs_per_s": n * (n - 1) / 2 / (best["triangle_ms"] * 1e-3)})
    return best


def realistic_workloads(engine, ctx, torch, synth, args):
    """Workloads next to the headline (never part of `value`): the headline's generator has substitutions only, which is the
    engine's best case -- no indels, so nearly every chunk is one run."""
    out = {}
    gold = os.path.join(ROOT, "tests", "golden", "genomes")
    recs = []
    if os.path.isdir(gold):
        recs = [_read_fasta_records(os.path.join(gold, n)) for n in sorted(os.listdir(gold))]
        r = _triangle_stats(engine, ctx, torch, [x[0] for x in recs], [x[1] for x in recs], 89.5)
        r["workload"] = "the 34 real Cutibacterium granulosum assemblies of the reference's GTDB test run (1-391 contigs), all 561 pairs, screen 89.5"
        out["real_34_genomes"] = r
    # the engine's ANI against the GENERATOR'S TRUTH over 86-99.95 % (skder_amd.synth.truth_recipe): the range the headline's
    # pairs sit in, where the golden skani tables (96.4-100 %, one species) say nothing
    try:
        rec = synth.truth_recipe(args.genome_len)
        truth = synth.true_identity_matrix(rec)
        layout = engine.BatchLayout(rec.rec_lens)
        d = torch.empty(layout.total_bytes, dtype=torch.uint8, device="cuda")
        ctx.synth_fill(d.data_ptr(), layout, rec.lineage, rec.params)
        sk = engine.Sketches(ctx)
        sk.sketch_batch(d.data_ptr(), layout)
        e = sk.triangle_rows(0, 1, args.screen)
        sk.close()
        del d
        out["ani_vs_truth"] = {"bins_true_ANI_pct": synth.ani_vs_truth(e, truth), "unit": "percentage points (engine - truth)",
                               "what": "%d synthetic genomes of %.1f Mb, one species, strain substitution rates 0-8 %%, no accessory segments: the true identity "
                                       "of each of the %d pairs is the fraction of equal bases. raw = chunk-level k-mer estimate (A/N)^(1/15) "
                                       "(skder_edge_t.ani_raw): unbiased. model = the table's ANI after the learned-ANI stand-in, fitted to skani's "
                                       "output on REAL genomes (clustered mutations): reads ~1.24 x the true divergence on iid substitutions"
                                       % (rec.n, args.genome_len / 1e6, rec.n * (rec.n - 1) // 2)}
    except Exception as ex:      # never lose the headline over an extra
        out["ani_vs_truth"] = {"error": str(ex)}
    # ... and from the other side: substitutions that CLUSTER per 1 kb window (synth.clustered_truth_family), truth known.  Divergence the
    # engine reads / true divergence: the raw estimate falls below 1 (intact k-mers survive in the quiet windows), the stand-in brings it back
    try:
        cl = {}
        for shape in (1.0, 0.3):
            bases, truth = synth.clustered_truth_family(2_000_000, shape=shape)
            layout = engine.BatchLayout([np.array([len(b)], np.uint32) for b in bases])
            d = torch.from_numpy(layout.pack_host(bases)).cuda()
            sk = engine.Sketches(ctx)
            sk.sketch_batch(d.data_ptr(), layout)
            e = sk.triangle_rows(0, 1, args.screen)
            sk.close()
            del d
            q = [(100.0 * (1.0 - truth[int(x["ref"]), int(x["query"])]), 1.0 - float(x["ani_raw"]), 1.0 - float(x["ani"])) for x in e]
            raw = [100.0 * r / t for t, r, m in q]
            mod = [100.0 * m / t for t, r, m in q]
            cl["gamma_shape_%g" % shape] = {"pairs": len(q), "true_divergence_pct": [round(min(t for t, _, _ in q), 3), round(max(t for t, _, _ in q), 3)],
                                            "raw_over_truth": [round(min(raw), 3), round(max(raw), 3)], "model_over_truth": [round(min(mod), 3), round(max(mod), 3)]}
        cl["what"] = ("7 genomes of 2 Mb, one ancestor, substitution rate of every 1 kb window x Gamma(shape, 1/shape): [min, max] over the 21 pairs of "
                      "(divergence read) / (true divergence); iid substitutions above: raw 1.0, model 1.24")
        out["ani_vs_truth"]["clustered"] = cl
    except Exception as ex:
        out["ani_vs_truth"]["clustered"] = {"error": str(ex)}
    # REAL genome structure at scale: every one of the 34 assemblies with args.real_derived descendants (substitutions, short
    # indels, inversions / translocations / deletions): one species, > 1,000 genomes, every pair chained -- what dereplicating
    # a well-sampled species looks like to the engine (contig ends, repeats, 10 % of the chunks on the unabridged path)
    if os.path.isdir(gold) and args.real_derived > 0:
        try:
            from concurrent.futures import ThreadPoolExecutor
            t0 = time.perf_counter()
            jobs = [(1000 * a + d, recs[a][1], recs[a][0]) for a in range(len(recs)) for d in range(args.real_derived)]
            with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
                fam = list(ex.map(lambda j: _real_descendant(*j), jobs))
            t_gen = time.perf_counter() - t0
            r = _triangle_stats(engine, ctx, torch, [g[1] for g in fam], [g[0] for g in fam], 80.0, steps=1)
            r["generate_s"] = t_gen
            r["workload"] = ("%d genomes: each of the reference's 34 real C. granulosum assemblies with %d host-generated descendants (0.02-3 %% substitutions, "
                             "one short indel per ~12 substitutions, up to 3 inversions / translocations / deletions of 0.5-20 kb, the assembly's own contig "
                             "structure); one species, every pair passes the screen and is chained" % (len(fam), args.real_derived))
            out["real_derived_%d" % len(fam)] = r
            del fam
        except Exception as ex:
            out["real_derived"] = {"error": repr(ex)}
    rng = np.random.RandomState(11)
    L = args.genome_len
    anc = np.frombuffer(b"ACGT", np.uint8)[rng.randint(0, 4, L)]
    fam = [_indel_descendant(rng, anc) for _ in range(args.indel_genomes)]
    r = _triangle_stats(engine, ctx, torch, [g[1] for g in fam], [g[0] for g in fam], args.screen)
    r["workload"] = ("%d host-generated descendants of one %.1f Mb ancestor: 0.05-4 %% substitutions and one short indel (geometric, mean 2.5 "
                     "bases) per ~12 substitutions, 1-30 records; all pairs chained" % (args.indel_genomes, L / 1e6))
    out["indel_%.0fMb" % (L / 1e6)] = r
    del fam
    # BASELINE.json config 5's shape on one GPU: species lengths uniform in 1-8 Mb
    n = args.mixed_genomes
    if n > 0:
        recipe = synth.make_recipe(n, len_range=(1_000_000, 8_000_000))
        batches, total = [], 0
        for b0 in range(0, n, 1250):
            gs = range(b0, min(b0 + 1250, n))
            layout = engine.BatchLayout([recipe.rec_lens[g] for g in gs])
            d = torch.empty(layout.total_bytes, dtype=torch.uint8, device="cuda")
            ctx.synth_fill(d.data_ptr(), layout, recipe.lineage[gs.start:gs.stop], recipe.params[gs.start:gs.stop])
            batches.append((layout, d))
            total += layout.total_bases
        best = None
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sk = engine.Sketches(ctx)
            sk.reserve(total // 120, total // 900)
            for layout, d in batches:
                sk.sketch_batch(d.data_ptr(), layout)
            edges = sk.triangle_rows(0, 1, args.screen, copy=False)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            c = ctx.counters()
            cur = {"ms_per_step": dt * 1e3, "edges": int(len(edges)), "chunks": int(c[0]), "slow_path_chunks": int(c[1])}
            sk.close()
            if best is None or cur["ms_per_step"] < best["ms_per_step"]:
                best = cur
        best.update({"genomes": n, "pairs": n * (n - 1) // 2, "bases": int(total), "pairs_per_s": n * (n - 1) / 2 / (best["ms_per_step"] * 1e-3),
                     "slow_path_fraction": best["slow_path_chunks"] / max(best["chunks"], 1),
                     "workload": "%d synthetic genomes, species lengths uniform in 1-8 Mb (BASELINE.json config 5's shape), one full step incl. sketching" % n})
        out["mixed_1_8Mb"] = best
        del batches
    return out


def n50_of_lengths(lens):
    """util.n50_calc's rule (util.py:686-724) on a list of record lengths: descending, half = int(sum / 2), first cumulative >= half"""
    ls = sorted((int(x) for x in lens), reverse=True)
    half, cum = int(sum(ls) / 2), 0
    for l in ls:
        cum += l
        if cum >= half:
            return l
    return 0


def low_mem_greedy_scale(engine, ctx, torch, synth, n, genome_len, device):
    """The reference's ONE published number (README.md:27): `skder -d low_mem_greedy` on > 20,000 Staphylococcus genomes took
    2.25 h on 20 threads (machine unspecified): skani sketch once, then one `skani search` process per representative
    (skder.py:95-134).  Here: the same loop (skder_amd.skder.lowMemGreedyDerep, -i 99.5 -f 50) over n synthetic genomes of
    BASELINE.json config 4's shape, the sketch database resident in HBM -- built from bases generated on the device (the
    FASTA ingest is priced separately under end_to_end) and, second, re-loaded from a sketch store on disk --, speculative
    search batches (the default) and search_batch=1 (one search, one TSV, one parse per representative, call for call)."""
    import shutil
    import tempfile
    from skder_amd.skder import Database, lowMemGreedyDerep
    t_all = time.perf_counter()
    recipe = synth.make_recipe(n, genome_len=genome_len)
    sk = engine.Sketches(ctx)
    total = sum(recipe.total_len(g) for g in range(n))
    sk.reserve(total // 120, total // 900)
    t0 = time.perf_counter()
    for b0 in range(0, n, 1250):
        gs = range(b0, min(b0 + 1250, n))
        layout = engine.BatchLayout([recipe.rec_lens[g] for g in gs])
        d = torch.empty(layout.total_bytes, dtype=torch.uint8, device="cuda")
        ctx.synth_fill(d.data_ptr(), layout, recipe.lineage[gs.start:gs.stop], recipe.params[gs.start:gs.stop])
        sk.sketch_batch(d.data_ptr(), layout)
        del d
    torch.cuda.synchronize()
    t_sketch = time.perf_counter() - t0
    tmp = tempfile.mkdtemp(prefix="skder_amd_lowmem_")
    out = {}
    try:
        paths = ["/synthetic/species%03d/g%05d.fasta" % (int(recipe.species[g]), g) for g in range(n)]
        n50 = [n50_of_lengths(recipe.rec_lens[g]) for g in range(n)]
        listing, n50_file = os.path.join(tmp, "listing.txt"), os.path.join(tmp, "Concatenated_N50.txt")
        open(listing, "w").write("".join(p + "\n" for p in paths))
        open(n50_file, "w").write("".join("%s\t%d\n" % kv for kv in zip(paths, n50)))
        t0 = time.perf_counter()
        db = Database.from_sketches(sk, paths, n50, device=device)
        t_db = time.perf_counter() - t0
        sk.close()

        def run(db, width, tag):
            ws = os.path.join(tmp, "ws_" + tag) + "/"
            os.makedirs(ws, exist_ok=True)
            res = os.path.join(ws, "skDER_Results.txt")
            t0 = time.perf_counter()
            lowMemGreedyDerep(listing, ws, n50_file, res, ws, 99.5, 50.0, None, search_batch=width, database=db)
            dt = time.perf_counter() - t0
            return dt, open(res).read()
        t_spec, reps_spec = run(db, 0, "spec")
        t_seq, reps_seq = run(db, 1, "seq")
        store = os.path.join(tmp, "sketches.skdr")
        store_s = load_s = t_store_run = None
        same_store = None
        if shutil.disk_usage(tmp).free > 20e9:
            t0 = time.perf_counter()
            db.save(store)
            store_s = time.perf_counter() - t0
            store_bytes = os.path.getsize(store)
            db.close()
            t0 = time.perf_counter()
            db = Database.load(store, device=device)
            load_s = time.perf_counter() - t0
            t_store_run, reps_store = run(db, 0, "store")
            same_store = reps_store == reps_spec
        db.close()
        reps = reps_spec.split()
        sp = {p: int(recipe.species[i]) for i, p in enumerate(paths)}
        out = {"genomes": n, "bases": int(total), "representatives": len(reps), "species_represented": len({sp[r] for r in reps}), "species": int(recipe.species.max()) + 1,
               "seconds_speculative_batches": t_spec, "seconds_one_search_per_representative": t_seq, "listings_identical": reps_spec == reps_seq,
               "database_from_resident_sketches_s": t_db, "sketching_s_incl_generating_the_bases": t_sketch,
               "store": None if store_s is None else {"write_s": store_s, "bytes": store_bytes, "load_and_index_s": load_s, "seconds_speculative_batches": t_store_run, "listing_identical": same_store},
               "reference_published": {"seconds": 2.25 * 3600, "what": "README.md:27: low_mem_greedy on > 20,000 Staphylococcus genomes (GTDB R220), 20 threads, machine unspecified; "
                                       "not a same-host, same-data comparison -- real genomes, FASTA ingest and skani's own arithmetic on one side, synthetic genomes already sketched on the other"},
               "total_s_of_this_leg": time.perf_counter() - t_all,
               "workload": "%d synthetic genomes x %.1f Mb (%d species x 10 strains x 10 isolates), lowMemGreedyDerep -i 99.5 -f 50 (skder.py:95-134) on one MI355X"
                           % (n, genome_len / 1e6, int(recipe.species.max()) + 1)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def one_species_database(engine, ctx, torch, synth, n, device, keep_bases=None):
    """n genomes of ONE species with real genome structure, sketched: the reference's 34 real C. granulosum assemblies and descendants
    of them generated on the device (skder_amd/csrc/descend.hip: substitutions 0.02 - 3 %, short indels, inversions / translocations /
    deletions, the assembly's own contigs).  Returns (sketch set, paths, n50, seconds).  keep_bases(batch_index, device tensor, layout):
    called per batch before it is dropped (the file-based variant writes FASTA from it)."""
    gold = os.path.join(ROOT, "tests", "golden", "genomes")
    names = sorted(os.listdir(gold))
    recs = [_read_fasta_records(os.path.join(gold, x)) for x in names]
    anc_layout = engine.BatchLayout([r[0] for r in recs])
    d_anc = torch.from_numpy(anc_layout.pack_host([r[1] for r in recs])).cuda()
    per = max((n - len(recs) + len(recs) - 1) // len(recs), 0)
    plan = synth.real_family_plan([r[0] for r in recs], per)
    # descendant k of every assembly before descendant k + 1 of any: a prefix of the plan is a balanced family
    order = np.argsort(np.arange(len(plan)) % max(per, 1), kind="stable")
    plan = plan[order][:max(n - len(recs), 0)]
    sk = engine.Sketches(ctx)
    sk.reserve(n * 18000, n * 2300)
    t0 = time.perf_counter()
    sk.sketch_batch(d_anc.data_ptr(), anc_layout)
    paths = ["/one_species/%s" % x for x in names]
    n50 = [n50_of_lengths(r[0]) for r in recs]
    if keep_bases:
        keep_bases(0, d_anc, anc_layout)
    step = 1000
    for b0 in range(0, len(plan), step):
        part = plan[b0:b0 + step]
        d, lay = ctx.descendants(d_anc.data_ptr(), anc_layout, part, torch)
        sk.sketch_batch(d.data_ptr(), lay)
        for g in range(len(part)):
            a, b = int(lay.genome_rec_begin[g]), int(lay.genome_rec_begin[g + 1])
            n50.append(n50_of_lengths(lay.rec_len[a:b]))
            paths.append("/one_species/%s.descendant%05d.fasta" % (names[int(part[g]["parent"])].split(".")[0], b0 + g))
        if keep_bases:
            keep_bases(1 + b0 // step, d, lay)
        del d
    torch.cuda.synchronize()
    return sk, paths, n50, time.perf_counter() - t0


def low_mem_greedy_one_species(engine, ctx, torch, synth, n, device, sequential=True):
    """The reference's published workload in its REAL SHAPE (README.md:27: `skder -d low_mem_greedy` on > 20,000 genomes of one genus,
    mostly one species, 2.25 h on 20 threads): lowMemGreedyDerep -i 99.5 -f 50 (skder.py:95-134) over n genomes of ONE species with
    real genome structure -- nearly every `search` passes the screen against thousands of genomes and chains them all, unlike the
    200-species synthetic leg where a search chains ~100 pairs."""
    import shutil
    import tempfile
    from skder_amd.skder import Database, lowMemGreedyDerep
    t_all = time.perf_counter()
    sk, paths, n50, t_sketch = one_species_database(engine, ctx, torch, synth, n, device)
    tmp = tempfile.mkdtemp(prefix="skder_amd_onesp_")
    try:
        listing, n50_file = os.path.join(tmp, "listing.txt"), os.path.join(tmp, "Concatenated_N50.txt")
        open(listing, "w").write("".join(p + "\n" for p in paths))
        open(n50_file, "w").write("".join("%s\t%d\n" % kv for kv in zip(paths, n50)))
        t0 = time.perf_counter()
        db = Database.from_sketches(sk, paths, n50, device=device)
        t_db = time.perf_counter() - t0
        sk.close()

        def run(width, tag):
            ws = os.path.join(tmp, "ws_" + tag) + "/"
            os.makedirs(ws, exist_ok=True)
            res = os.path.join(ws, "skDER_Results.txt")
            t0 = time.perf_counter()
            lowMemGreedyDerep(listing, ws, n50_file, res, ws, 99.5, 50.0, None, search_batch=width, database=db)
            return time.perf_counter() - t0, open(res).read(), dict(getattr(lowMemGreedyDerep, "last_stats", {}))
        t_spec, reps_spec, st = run(0, "spec")
        t_seq, reps_seq = None, reps_spec
        if sequential:
            t_seq, reps_seq, _ = run(1, "seq")
        db.close()
        nrep = len(reps_spec.split())
        return {"genomes": len(paths), "representatives": nrep, "seconds_speculative_batches": t_spec, "seconds_one_search_per_representative": t_seq,
                "listings_identical": (reps_spec == reps_seq) if sequential else None, "searches": st.get("searches"), "search_batches": st.get("batches"),
                "rows_per_search": (st.get("rows", 0) / max(st.get("searches", 1), 1)), "rows_total": st.get("rows"),
                "seconds_inside_search_batch": st.get("search_s"), "seconds_rows_pass": st.get("rows_pass_s"),
                "sketching_s_incl_generating_the_bases": t_sketch, "database_from_resident_sketches_s": t_db,
                "reference_published": {"seconds": 2.25 * 3600, "what": "README.md:27: low_mem_greedy on > 20,000 Staphylococcus genomes (GTDB R220), 20 threads, machine "
                                        "unspecified; other genomes, other hardware, FASTA ingest and skani's own arithmetic included there -- quoted, not compared"},
                "total_s_of_this_leg": time.perf_counter() - t_all,
                "workload": "%d genomes of ONE species: the reference's 34 real C. granulosum assemblies + descendants generated on the device (0.02-3 %% substitutions, "
                            "short indels, inversions / translocations / deletions, the assemblies' own contigs); lowMemGreedyDerep -i 99.5 -f 50 on one MI355X, sketches resident"
                            % len(paths)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def cpu_baseline_files(tmp, paths, threads):
    """oracle (CPU restatement, OpenMP) on the sample files, wall clock on `threads` host threads.  MEASURED: one `triangle` over all the
    files -- every pair screened, every screened pair chained.  Beside it three rates for the extrapolation to the full workload:
      per genome   : read + sketch, from a triangle over the first files whose 101 % screen lets no pair through;
      per pair     : the pairwise marker screen, timed on one thread over cross-genome pairs and divided
                     by `threads` (perfect scaling assumed: the optimistic choice for the CPU);
      per chained  : the full triangle's remaining time per pair that passes the screen."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
    p = oracle_py.default_params()

    def run(sub, screen):
        listing = os.path.join(tmp, "cpu_listing.txt")
        open(listing, "w").write("".join(q + "\n" for q in sub))
        out = os.path.join(tmp, "cpu_edges.tsv")
        t0 = time.perf_counter()
        oracle_py.triangle(listing, 0.0, screen, threads, out, p)
        dt = time.perf_counter() - t0
        return dt, sum(1 for _ in open(out)) - 1

    n = len(paths)
    t_full, chained = run(paths, 80.0)
    n_load = min(n, 512)
    t_load, _ = run(paths[:n_load], 101.0)
    k = min(n, 24)
    gs = [oracle_py.Genome.load(q, p) for q in paths[:k]]
    t0 = time.perf_counter()
    npair = 0
    for rep in range(4):
        for i in range(k):
            for j in range(i + 1, k):
                oracle_py.screen(gs[i], gs[j], 80.0, p)
                npair += 1
    t_screen = (time.perf_counter() - t0) / max(npair, 1)
    per_pair = t_screen / threads
    per_genome = t_load / n_load
    per_chained = max(t_full - per_genome * n - per_pair * (n * (n - 1) // 2), 0.0) / max(chained, 1)
    measured = {"files": n, "pairs": n * (n - 1) // 2, "chained_pairs": chained, "wall_s": t_full, "pairs_per_s": (n * (n - 1) // 2) / t_full,
                "what": "one `triangle` of the oracle over the sample files, every pair screened, every screened pair chained: MEASURED wall clock"}
    return per_genome, per_pair, per_chained, chained, t_load + t_full + t_screen * npair, measured


def parity_sample(recipe, edges, n_pairs, seed=5):
    """The headline's own output against the oracle, at the metric's own size: n_pairs random edge records of the last step
    -- integer sums and doubles -- against oracle_py.pair on the same two genomes (bases regenerated on the host by the numpy
    statement of the generator).  After the timed region; the oracle is the checker here, never the thing measured."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
    from skder_amd import synth
    rng = np.random.RandomState(seed)
    pick = rng.choice(len(edges), size=min(n_pairs, len(edges)), replace=False) if len(edges) else []
    p = oracle_py.default_params()
    cache, bad, t0 = {}, [], time.perf_counter()

    def genome(g):
        if g not in cache:
            cache[g] = oracle_py.Genome.from_bases(synth.bases_numpy(recipe, g), recipe.rec_lens[g], p)
        return cache[g]
    for i in pick:
        e = edges[int(i)]
        a, b = int(e["ref"]), int(e["query"])
        r = oracle_py.pair(genome(a), genome(b), p)
        same = (int(e["sum_anchors"]) == r.sum_anchors and int(e["sum_seeds"]) == r.sum_seeds and int(e["cell_seeds"]) == r.cell_seeds
                and int(e["aligned_bases"]) == r.aligned_bases and float(e["ani"]) == r.ani and float(e["ani_raw"]) == r.ani_raw
                and float(e["af_ref"]) == r.af_ref and float(e["af_query"]) == r.af_query)
        if not same:
            bad.append([a, b])
    return {"pairs": int(len(pick)), "mismatches": len(bad), "mismatched_pairs": bad[:5], "seconds": time.perf_counter() - t0,
            "what": "random edge records of the headline's last step (the shipped two-queue path) against oracle_py.pair on the same genomes: "
                    "anchors, seeds, cell seeds, aligned bases, ANI (raw and model) and both aligned fractions, bit for bit"}


def cpu_baseline_skani(tmp, paths, threads):
    """The reference engine itself, when the box has it (oracle/skani_ref.py): `skani triangle -t threads`
    on the sample files, wall-clocked, next to a run whose screen lets nothing through but identical genomes (-s 100),
    which prices read + sketch per genome; the difference per chained pair prices the rest.  None when skani is absent."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import skani_ref
    if not skani_ref.find():
        return None
    listing = os.path.join(tmp, "skani_listing.txt")
    open(listing, "w").write("".join(q + "\n" for q in paths))
    full = skani_ref.triangle(listing, os.path.join(tmp, "skani_full.tsv"), 50.0, 80.0, threads)
    load = skani_ref.triangle(listing, os.path.join(tmp, "skani_load.tsv"), 50.0, 100.0, threads)
    n = len(paths)
    chained = max(full["rows"], 1)
    return {"per_genome": load["seconds"] / n, "per_chained": max(full["seconds"] - load["seconds"], 0.0) / chained, "chained": full["rows"],
            "seconds": full["seconds"], "sample_pairs_per_s": n * (n - 1) / 2 / full["seconds"], "version": full["version"], "command": full["command"]}


def end_to_end_legs(out, tmp, all_paths, sizes, args, dev, total_bases, ms_per_step):
    """second clock: listing file -> TSV on disk through the drop-in entry point, on bounded samples; never part of
    `value`.  Two nested samples (the first e2e_genomes files, and four times as many): the call has a fixed part (context, pinned
    staging buffers, first-use allocations, the triangle of a small set) that a 0.8 GB sample cannot amortise; the RATE of the
    ingest is the slope between the two, and the extrapolation uses intercept + slope.  Fills out["end_to_end"] and ["end_to_end_gz"]."""
    n_small = min(args.e2e_genomes, len(all_paths))
    big_paths = all_paths[:min(4 * args.e2e_genomes, len(all_paths))]
    big_sizes = sizes[:len(big_paths)]
    paths, nbytes = big_paths[:n_small], sum(big_sizes[:n_small])

    def two_point(small_paths, small_bytes, big, big_bytes):
        end_to_end_sample(tmp, small_paths[:8], sum(big_sizes[:8]), dev)      # code objects and the context's first-use allocations exist after this
        cold = end_to_end_sample(tmp, small_paths, small_bytes, dev)      # pins the staging buffers at their working size (once per process)
        a = end_to_end_sample(tmp, small_paths, small_bytes, dev)
        a["first_call_s"] = cold["seconds"]
        if len(big) <= len(small_paths):
            return a
        b = min((end_to_end_sample(tmp, big, big_bytes, dev) for _ in range(2)), key=lambda r: r["seconds"])
        if b["seconds"] <= a["seconds"]:         # the big sample was not slower (noise): no slope to extrapolate from
            b["small_sample"] = a
            return a if a["seconds"] < b["seconds"] else b
        slope = (b["fasta_bytes"] - a["fasta_bytes"]) / (b["seconds"] - a["seconds"])
        b["small_sample"] = a
        b["marginal_MB_per_s"] = slope / 1e6
        b["fixed_s"] = max(a["seconds"] - a["fasta_bytes"] / slope, 0.0) + max(cold["seconds"] - a["seconds"], 0.0)
        return b
    e = two_point(paths, nbytes, big_paths, sum(big_sizes))
    if "marginal_MB_per_s" in e:
        e["extrapolated_full_workload_s"] = e["fixed_s"] + total_bases * 1.0125 / (e["marginal_MB_per_s"] * 1e6)
    else:
        e["extrapolated_full_workload_s"] = total_bases * 1.0125 / (e["fasta_bytes"] / e["seconds"]) + ms_per_step * 1e-3
    try:
        # the same call with the files' pages dropped from the page cache first: what the storage of this box gives
        fs = evict_from_page_cache(big_paths)
        cold = end_to_end_sample(tmp, big_paths, sum(big_sizes), dev)
        e["cold_page_cache"] = {"seconds": cold["seconds"], "ingest_MB_per_s": cold["ingest_MB_per_s"], "file_system": fs,
                                "how": "fsync + posix_fadvise(DONTNEED) on every file, then the call once"}
    except Exception as ex:
        e["cold_page_cache"] = {"error": repr(ex)}
    e["sample"] = ("skder_amd_triangle_n50 on %d FASTA files of the workload (every second genome; page cache hot): read, PCIe copy, FASTA parse on the device, N50, "
                   "sketch, index, screen, chain, TSV. ingest_MB_per_s = bytes / seconds of the whole call; marginal_MB_per_s = the slope "
                   "between this sample and its first %d files (small_sample; both with the staging buffers of an earlier call), i.e. the ingest "
                   "pipeline's rate without the call's fixed part; fixed_s = the small sample's intercept + what its FIRST call in the process "
                   "took longer (pinning the staging buffers: first_call_s); extrapolation = fixed_s + the full workload's FASTA bytes at the marginal rate"
                   % (e["genomes"], len(paths)))
    out["end_to_end"] = e
    # the same samples as .fasta.gz: one zlib stream per file, inflated on the host threads beside the parser
    gz_all, _ = gzip_sample_files(big_paths)
    gz_sizes = [os.path.getsize(q) for q in gz_all]
    eg = two_point(gz_all[:len(paths)], nbytes, gz_all, sum(big_sizes))
    eg["gz_bytes"] = sum(gz_sizes[:eg["genomes"]])
    eg["gz_MB_per_s"] = eg["gz_bytes"] / eg["seconds"] / 1e6
    if "marginal_MB_per_s" in eg:
        eg["extrapolated_full_workload_s"] = eg["fixed_s"] + total_bases * 1.0125 / (eg["marginal_MB_per_s"] * 1e6)
    eg["sample"] = ("the same files gzip-compressed (level 1, %.2f x): skder_amd_triangle_n50 from .fasta.gz; ingest_MB_per_s and marginal_MB_per_s count "
                    "uncompressed FASTA bytes, gz_MB_per_s compressed ones; page cache hot" % (sum(big_sizes) / max(sum(gz_sizes), 1)))
    out["end_to_end_gz"] = eg
    for q in gz_all:
        os.remove(q)


def cpu_baseline(tmp, all_paths, N, pairs, n_chained):
    """the `cpu_baseline` object of the line: the reference engine itself when the box has it (kind "reference"), else the oracle
    (kind "port").  `value` is MEASURED -- the sample's pairs over the wall clock of one triangle on the sample, whose share of chained
    pairs is the full workload's; the extrapolation to the full workload from the per-stage rates is a side field."""
    threads = max(1, min(32, granted_cpus()))
    n = len(all_paths)
    sample_what = ("every second genome of the workload: %d FASTA files, %d pairs -- every species of the workload with half of its strains, so the "
                   "share of pairs that are chained equals the full workload's" % (n, n * (n - 1) // 2))
    sk = cpu_baseline_skani(tmp, all_paths, granted_cpus())
    if sk is not None:
        est = N * sk["per_genome"] + n_chained * sk["per_chained"]
        return {"value": sk["sample_pairs_per_s"], "unit": "genome-pairs/s", "cores": granted_cpus(), "kind": "reference", "skani_version": sk["version"],
                "value_is": "measured: the sample's pairs / the wall clock of `skani triangle` on the sample", "extrapolated_full_workload_pairs_per_s": pairs / est,
                "sample": "%s; %s, wall clock %.1f s, %d rows; a -s 100 run prices read + sketch at %.4f s/genome, the rest is %.5f s per chained pair"
                          % (sample_what, sk["command"], sk["seconds"], sk["chained"], sk["per_genome"], sk["per_chained"])}
    pg, pp, pc, chained, spent, measured = cpu_baseline_files(tmp, all_paths, threads)
    est = N * pg + pairs * pp + n_chained * pc
    return {"value": measured["pairs_per_s"], "unit": "genome-pairs/s", "cores": threads, "kind": "port", "measured_on_sample": measured,
            "value_is": "measured: the sample's pairs / the wall clock of one oracle `triangle` over the sample's files (read, sketch, every pair screened, "
                        "every screened pair chained); the sample has %.2f x the full workload's share of chained pairs" 
                        % ((measured["chained_pairs"] / max(measured["pairs"], 1)) / max(n_chained / max(pairs, 1), 1e-12)),
            "extrapolated_full_workload_pairs_per_s": pairs / est,
            "extrapolation": "genomes x %.4f s (read + sketch) + pairs x %.2e s (marker screen) + chained pairs x %.5f s, the three rates measured on the sample: "
                             "%d genomes, %d pairs, %d chained pairs" % (pg, pp, pc, N, pairs, int(n_chained)),
            "skani": "unavailable on this host (oracle/skani_ref.py looked for it on PATH and in the usual conda locations): the repo's own CPU restatement is timed instead",
            "sample": "%s; oracle (CPU restatement, OpenMP, %d threads, wall clock), %.1f s of wall time spent in this leg" % (sample_what, threads, spent)}


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: N ranks of this script under torch.distributed.run (one rank per GPU over RCCL),
    started as a child process BEFORE torch or the library is imported here; the rendezvous is on 127.0.0.1 at a free port."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)     # the first steps of the first process on a box run a few ms slower
    ap.add_argument("--genomes", type=int, default=5000)
    ap.add_argument("--genome-len", type=int, default=3_000_000)
    ap.add_argument("--len-range", type=int, nargs=2, default=None, metavar=("LO", "HI"),
                    help="species lengths uniform in [LO, HI] (BASELINE config 5: 1000000 8000000) instead of --genome-len +-5 %%")
    ap.add_argument("--screen", type=float, default=80.0)
    ap.add_argument("--batch-genomes", type=int, default=2500, help="genomes per resident input batch (one sketch call each)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--e2e-genomes", type=int, default=256, help="genomes in the file-based end-to-end sample (0: skip)")
    ap.add_argument("--cpu-sample-genomes", type=int, default=2500, help="genomes of the CPU baseline's sample: every second genome of the workload, at most this many")
    ap.add_argument("--dump-edges", default=None, metavar="FILE.npy", help="rank 0 writes the last step's edge records, sorted by (ref, query)")
    ap.add_argument("--no-realistic", action="store_true", help="skip the extra workloads (real genomes, indels, mixed sizes)")
    ap.add_argument("--indel-genomes", type=int, default=48, help="genomes of the host-generated indel family")
    ap.add_argument("--mixed-genomes", type=int, default=5000, help="genomes of the mixed 1-8 Mb extra workload (0: skip)")
    ap.add_argument("--real-derived", type=int, default=30, help="descendants per real assembly in the real-structure workload (34 x this many genomes; 0: skip)")
    ap.add_argument("--one-species-genomes", type=int, default=5000, help="genomes of the one-species low_mem_greedy leg (README.md:27's workload in its real shape; "
                                                                            "profiles/run/r4_one_species.py runs 20000; 0: skip)")
    ap.add_argument("--parity-pairs", type=int, default=20, help="edges of the headline's last step checked against the CPU oracle after the timed region (0: skip)")
    ap.add_argument("--low-mem-genomes", type=int, default=20000, help="genomes of the low_mem_greedy leg (README.md:27's workload shape: 20000; 0: skip)")
    args = ap.parse_args()

    # --gpus N is the contract's flag.  Under a launcher (WORLD_SIZE set) it must agree with the launcher's world size; without one and
    # with N > 1 this process starts N ranks itself, as a CHILD (never an exec: nothing here has touched the GPU yet, and the child's
    # stdout is this process's, so the JSON line stays the last line), and returns the child's exit code.
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d does not match the launcher's WORLD_SIZE=%d (run `python bench.py --gpus N` without a launcher, "
                 "or pass the same N to both)" % (args.gpus, world))

    import torch
    import torch.distributed as dist
    from skder_amd import engine, multigpu, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # SKDER_AMD_DIST_BACKEND=gloo lets several ranks share one GPU (functional check of the N>1 path on
    # a 1-GPU box: the exchange then goes through host memory); the default is RCCL, one GPU per rank
    backend = os.environ.get("SKDER_AMD_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev = local_rank % max(ndev, 1)
    # SKDER_AMD_FORCE_DIST=1 (under torch.distributed.run with one process) drives the whole N > 1 code
    # path -- RCCL process group, sketch all-gather on device tensors, edge gather -- on a single GPU
    dist_on = world > 1 or (os.environ.get("SKDER_AMD_FORCE_DIST") == "1" and "MASTER_ADDR" in os.environ)
    if dist_on:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            if world > max(ndev, 0):
                sys.exit("bench.py: --gpus %d but this node shows %d GPU(s): RCCL needs one GPU per rank (SKDER_AMD_DIST_BACKEND=gloo lets "
                         "ranks share a GPU for a functional check)" % (world, ndev))
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend)
    torch.cuda.set_device(dev)
    ctx = engine.Context(dev)

    N = args.genomes
    recipe = synth.make_recipe(N, genome_len=args.genome_len, len_range=args.len_range)
    mine = multigpu.partition(N, world)[rank]
    # inputs: generated on the device, resident in HBM before the timed region
    batches = []
    for b0 in range(mine.start, mine.stop, args.batch_genomes):
        gs = range(b0, min(b0 + args.batch_genomes, mine.stop))
        layout = engine.BatchLayout([recipe.rec_lens[g] for g in gs])
        d = torch.empty(layout.total_bytes, dtype=torch.uint8, device="cuda")
        ctx.synth_fill(d.data_ptr(), layout, recipe.lineage[gs.start:gs.stop], recipe.params[gs.start:gs.stop])
        batches.append((layout, d))
    total_bases = sum(l.total_bases for l, _ in batches)
    torch.cuda.synchronize()

    wall = {"sketch": 0.0, "exchange": 0.0, "triangle": 0.0, "gather": 0.0}
    stage = {}          # N > 1: this rank's wall time per stage of the sharded triangle, summed over the timed steps

    def step():
        tm = np.zeros(8)
        t0 = time.perf_counter()
        sk = engine.Sketches(ctx)
        sk.reserve(total_bases // 120, total_bases // 900)     # densities are 1/125 and 1/1000: a few % of slack
        for layout, d in batches:
            sk.sketch_batch(d.data_ptr(), layout)
            t = ctx.timing()
            tm[0] += t[0]; tm[1] += t[1]
        t1 = time.perf_counter()
        by_components = dist_on and os.environ.get("SKDER_AMD_EXCHANGE") == "components"
        if dist_on and not by_components:
            raw = multigpu.exchange_raw(multigpu.raw_from_sketches(sk), staging="cpu" if backend != "nccl" else None, parts=True)
            sk.close()
            sk = multigpu.sketches_from_raw(ctx, raw)
        t2 = time.perf_counter()
        # no explicit sk.index(): triangle_rows builds the seed index itself, on a second stream beside the marker screen
        if by_components:
            # markers to everyone, every genome's seeds only to the rank that owns its connected component of candidate pairs
            edges = multigpu.triangle_by_components(ctx, sk, mine.start, N, rank, world, args.screen, staging="cpu" if backend != "nccl" else None)
            step.exchange_stats = dict(multigpu.triangle_by_components.last_stats)
        elif dist_on:
            # index only the genomes this rank owns, screen its rows, chain the pairs that probe its genomes
            edges = multigpu.triangle_sharded(sk, rank, world, args.screen, copy=False)
        else:
            edges = sk.triangle_rows(rank, world, args.screen, copy=False)   # a view of the library's host buffer
        t4 = time.perf_counter()
        wall["sketch"] += t1 - t0; wall["exchange"] += t2 - t1; wall["triangle"] += t4 - t2
        t = ctx.timing()
        tm[2:] = t[2:]
        step.index_ms = ctx.index_ms()
        step.runs_ms = ctx.runs_ms()
        step.counters = ctx.counters()
        if dist_on:
            t5 = time.perf_counter()
            edges = multigpu.gather_edges(edges, copy=False)
            wall["gather"] += time.perf_counter() - t5
            for k, v in getattr(multigpu.triangle_by_components if by_components else multigpu.triangle_sharded, "last_stage_ms", {}).items():
                stage[k] = stage.get(k, 0.0) + v
        sk.close()
        return edges, tm

    def sync():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    sync()
    for k in wall:
        wall[k] = 0.0
    stage.clear()
    t0 = time.perf_counter()
    tms = []
    for _ in range(args.steps):
        edges, tm = step()
        tms.append(tm)
    sync()
    dt = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    host_wall_ms = {k: 1e3 * v / args.steps for k, v in wall.items()}      # of the timed steps only (the extra steps below add to wall[])
    # every rank's wall time per stage and step (N > 1: so that the first scaling curve can be read: which stage does not shrink)
    my_stages = {"sketch": 1e3 * wall["sketch"] / args.steps, "exchange": 1e3 * wall["exchange"] / args.steps}
    my_stages.update({k: v / args.steps for k, v in stage.items()})
    my_stages["gather"] = 1e3 * wall["gather"] / args.steps
    per_rank = [my_stages]
    if dist_on:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, my_stages)
    pairs = N * (N - 1) // 2
    tm = np.mean(tms, axis=0)
    # In the timed region the chaining batches alternate between two queues and overlap, so the event bracket
    # of a chain-stage kernel there includes the chip time it shared with the other queue.  Per-kernel durations of that
    # stage are therefore taken from ONE_QUEUE_STEPS extra steps (their mean), outside the timed region, with all batches on one queue; the sketch
    # kernel (the longest, never overlapped) keeps the timed region's figure.
    overlapped = {"join_probe_kernel": float(step.counters[2]) / 1000.0, "run_extract_kernel": float(step.runs_ms),
                  "chain_single_kernel+chain_runs_kernel": float(tm[3]), "chain_slow_path": float(tm[4]), "finalize": float(tm[5])}
    one_queue_ms, one_queue_steps_ms = None, None
    if not dist_on and os.environ.get("SKDER_AMD_QUEUES") is None:
        # one discarded step after the queue count changes (the work buffers are laid out again), then ONE_QUEUE_STEPS steps timed one
        # by one; the figures are the MEAN over those steps, and every step's time is in the line so that an outlier can be seen
        os.environ["SKDER_AMD_QUEUES"] = "1"
        step()
        torch.cuda.synchronize()
        tq, one_queue_steps_ms = [], []
        for _ in range(ONE_QUEUE_STEPS):
            t1q = time.perf_counter()
            _, t_ = step()
            torch.cuda.synchronize()
            one_queue_steps_ms.append((time.perf_counter() - t1q) * 1e3)
            tq.append((t_, np.array(step.counters, copy=True), step.runs_ms, step.index_ms))
        del os.environ["SKDER_AMD_QUEUES"]
        # the MEAN of the one-queue steps, like ms_per_step is the mean of the timed steps: the same statistic on both sides of every ratio
        one_queue_ms = float(np.mean(one_queue_steps_ms))
        tm[2:6] = np.mean([q[0][2:6] for q in tq], axis=0)
        edges, _ = step()          # back on the shipped two queues: the edges checked below are those of the shipped path
        torch.cuda.synchronize()
        step.counters = np.mean([q[1].astype(np.float64) for q in tq], axis=0)
        step.runs_ms = float(np.mean([q[2] for q in tq]))
        step.index_ms = float(np.mean([q[3] for q in tq]))
    n_chained_all = float(tm[6])
    if dist_on:
        t = torch.tensor([tm[6]], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        n_chained_all = float(t.item())
    # N > 1: a few steps of the OTHER exchange as well, outside the timed region (never part of `value`): the first multi-GPU run of this
    # line then prices both -- replicate (raw sketches all-gathered) and components (markers all-gathered, seeds to their component's owner)
    other_exchange = None
    if dist_on and world > 1 and os.environ.get("SKDER_AMD_OTHER_EXCHANGE") == "1":       # opt-in: a collective that hangs would cost the whole line
        was = os.environ.get("SKDER_AMD_EXCHANGE")
        other = "replicate" if was == "components" else "components"
        kept = (np.array(step.counters, copy=True), step.runs_ms, step.index_ms, getattr(step, "exchange_stats", None))
        try:
            os.environ["SKDER_AMD_EXCHANGE"] = other
            step()
            sync()
            t0o = time.perf_counter()
            for _ in range(3):
                e_other, _ = step()
            sync()
            dto = torch.tensor([time.perf_counter() - t0o], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(dto, op=dist.ReduceOp.MAX)
            other_exchange = {"mode": other, "ms_per_step": float(dto.item()) / 3 * 1e3, "steps": 3, "edges": int(len(e_other)),
                              "stats_rank0": dict(getattr(step, "exchange_stats", {})) if other == "components" else None}
        except Exception as ex:          # the extra measurement must never cost the line
            other_exchange = {"mode": other, "error": repr(ex)}
        finally:
            if was is None:
                os.environ.pop("SKDER_AMD_EXCHANGE", None)
            else:
                os.environ["SKDER_AMD_EXCHANGE"] = was
            step.counters, step.runs_ms, step.index_ms = kept[:3]      # the line's per-kernel figures are those of the timed mode
            if kept[3] is not None:
                step.exchange_stats = kept[3]

    # the last collective is behind us: the ranks part here, and rank 0 finishes the line (and its CPU legs) on its own
    rccl_info = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "devices_visible": ndev} if dist_on else None
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and args.dump_edges:
        e = np.array(edges, copy=True)
        np.save(args.dump_edges, e[np.lexsort((e["query"], e["ref"]))])
    if rank == 0:
        # dominant kernel and its roofline (algorithmic bytes, DESIGN.md "Kernels")
        n_chained, n_anchors = tm[6], tm[7]
        sketch_bytes = total_bases * (1.0 + 8.0 / 125 + 8.0 / 1000)        # this rank's sketch kernel launches
        seeds_per_genome = args.genome_len / 125.0
        if args.len_range is not None:
            # mixed lengths: the chained pairs are the within-species pairs, so weight each species' length by its pair count
            ns = np.bincount(recipe.species)
            sl = np.array([recipe.total_len(int(np.argmax(recipe.species == s))) for s in range(len(ns))], np.float64)
            w = ns * (ns - 1) / 2.0
            seeds_per_genome = float((w * sl).sum() / max(w.sum(), 1.0)) / 125.0
        # join_probe_kernel: per chained pair the chunked genome's position-ordered k-mers are read once
        # (4 B per seed), one hit word per seed is written (4 B) and the matched position is gathered for
        # about 70 % of the seeds (4 B); the probed genome's index is staged in LDS once per <= 8 pairs;
        # run_extract_kernel: per chained pair the hit word, the position and the chunk-start flag of every seed are
        # read (9 B), the run records written are two orders of magnitude fewer;
        # chain_single_kernel (+ chain_runs_kernel for the 9 % of chunks it leaves): per chunk the first-record index (4 B),
        # on average two 32-byte run records and their closing record, the chunk's state (4 B) and its chain (32 B)
        join_bytes = n_chained * (11.0 * seeds_per_genome)
        runs_bytes = n_chained * (9.0 * seeds_per_genome)
        chain_bytes = float(step.counters[0]) * (4.0 + 3 * 32.0 + 4.0 + 32.0)
        join_ms = float(step.counters[2]) / 1000.0
        cand = {"sketch_tiles_kernel": (tm[0], sketch_bytes), "join_probe_kernel": (join_ms, join_bytes),
                "run_extract_kernel": (step.runs_ms, runs_bytes), "chain_single_kernel+chain_runs_kernel": (tm[3], chain_bytes)}
        # what actually limits each kernel (SQ counters under profiles/): none of them is at the HBM roof
        limiter = {"sketch_tiles_kernel": "VALU issue: two mm_hash64 per position; the body costs 93 ns per position and wavefront against 107 for the compiler's "
                                          "instruction selection (profiles/round3_sketch_body.json), HBM traffic = 1.07 x algorithmic",
                   "join_probe_kernel": "instruction issue: ~118 wavefront instructions per 64 probes (79 VALU, 25 SALU) at two workgroups per CU; "
                                        "LDS bank conflicts 7 % of LDS cycles (profiles/round4_pmc_join_probe_kernel.txt: the kernel is unchanged since)",
                   "run_extract_kernel": "HBM: 9 B per seed at ~5 TB/s",
                   "chain_single_kernel+chain_runs_kernel": "memory latency: a dozen dependent loads per wavefront"}
        bound_of = {"sketch_tiles_kernel": "valu-issue", "join_probe_kernel": "valu-issue", "run_extract_kernel": "hbm",
                    "chain_single_kernel+chain_runs_kernel": "latency"}
        dom = max(cand, key=lambda k: cand[k][0])
        dms, dbytes = cand[dom]
        achieved = dbytes / (dms * 1e-3) / 1e9 if dms > 0 else 0.0
        # the same step priced with SURVEY.md 8(d)'s layout-independent figures (16-byte seed records): sketch
        # 1.136 B per base over the sketch stage, chain 16 B x (both genomes' seeds) per chained pair over
        # join + chaining + finalize; the per-kernel figure above uses this build's smaller records instead
        def gbs(nbytes, ms):
            return {"bytes": float(nbytes), "ms": float(ms), "GB/s": float(nbytes / (ms * 1e-3) / 1e9) if ms > 0 else 0.0}
        s8_sketch = gbs(total_bases * (1.0 + 16.0 / 125 + 8.0 / 1000), tm[0] + tm[1])
        s8_chain = gbs(n_chained * 16.0 * 2.0 * seeds_per_genome, join_ms + step.runs_ms + tm[3] + tm[4] + tm[5])
        s8_all = gbs(s8_sketch["bytes"] + s8_chain["bytes"], ms_per_step)          # over the whole step's wall time
        # measured HBM traffic of the dominant kernel per step (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
        # separate passes, PMC_TRAFFIC; only valid for the default workload on 1 GPU)
        traffic, traffic_source, step_traffic = None, None, None
        if N == 5000 and world == 1 and args.len_range is None and args.genome_len == 3_000_000:
            traffic, traffic_source, step_traffic = pmc_traffic(dom.split("+")[0])
        # the VALU-bound kernel against its own issue floor: the inner body alone (profiles/calib/sketch_body_bench.hip, the shipped
        # instruction selection, ns per position and wavefront per SIMD) over what the whole kernel takes per position and wavefront
        ns_pw = float(tm[0] * 1e6 * 1024 / (total_bases / 64.0)) if total_bases else 0.0
        out = {
            "metric": "genome-pairs ANI/sec on 5k x 3Mb synthetic", "value": pairs / (ms_per_step * 1e-3),
            "unit": "genome-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u64 hash / i32 chaining / f64 ANI", "data": "synthetic",
            "rccl": rccl_info,
            "config": {"workload": "%d synthetic genomes x %s Mb (%d species x 10 strains x 10 isolates), triangle, screen %.0f"
                       % (N, "%.1f" % (args.genome_len / 1e6) if args.len_range is None else
                          "%.1f-%.1f" % (args.len_range[0] / 1e6, args.len_range[1] / 1e6), max(1, N // 100), args.screen), "genomes": N, "pairs": pairs,
                       "chained_pairs": int(n_chained_all), "edges": int(len(edges)),
                       "chunks": int(step.counters[0]), "slow_path_chunks": int(step.counters[1]),
                       "parallelism": "rows%d" % world},
            # achieved / peak / frac are the HBM figures the contract asks for (algorithmic bytes over the kernel's time against 8 TB/s);
            # `bound` says what actually limits the dominant kernel -- when that is instruction issue, `issue` prices the kernel
            # against the measured issue time of its own inner body (profiles/calib/sketch_body_bench.hip)
            "roofline": {"bound": bound_of[dom], "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "valu_frac": (SKETCH_BODY_NS / ns_pw if (dom == "sketch_tiles_kernel" and ns_pw > 0) else None),
                         "issue": ({"ns_per_position_wavefront_per_simd": ns_pw, "body_alone_ns": SKETCH_BODY_NS,
                                    "compiler_selected_body_ns": 107.4, "source": "profiles/round3_sketch_body.json (256 CUs x 4 SIMDs)",
                                    "valu_frac_is": "body_alone_ns / ns_per_position_wavefront_per_simd: the kernel against the measured issue time of "
                                                    "its own inner loop -- the yardstick for a kernel whose HBM fraction says nothing"}
                                   if dom == "sketch_tiles_kernel" else None),
                         "step_traffic_bytes": step_traffic,
                         "step_traffic_GBs": (step_traffic / (ms_per_step * 1e-3) / 1e9 if step_traffic else None),
                         "step_traffic_frac_of_peak": (step_traffic / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS if step_traffic else None),
                         "limited_by": limiter[dom], "algorithmic_bytes": dbytes,
                         "survey_8d": {"sketch": s8_sketch, "chain": s8_chain, "step": s8_all},
                         "kernel_ms": {k: float(v[0]) for k, v in cand.items()},
                         "kernel_ms_note": "chain-stage kernels (join, run extraction, sieve + run loop, slow path, finalize): HIP-event "
                                           "durations of three extra steps (mean) with all batches on ONE queue; in the timed region the batches "
                                           "alternate between two queues and overlap, and their event brackets (kernel_ms_two_queues) "
                                           "include the wait for the other queue's share of the chip",
                         "kernel_ms_two_queues": overlapped, "ms_per_step_one_queue": one_queue_ms, "one_queue_steps_ms": one_queue_steps_ms,
                         "host_wall_ms": host_wall_ms,
                         "per_rank_stage_ms": per_rank,
                         "kernel_GBs": {k: float(v[1] / (v[0] * 1e-3) / 1e9) if v[0] > 0 else 0.0 for k, v in cand.items()},
                         "other_ms": {"sketch_post": float(tm[1]), "index_beside_screen": float(step.index_ms), "screen": float(tm[2]), "chain_slow_path": float(tm[4]),
                                      "finalize": float(tm[5])}},
        }
        if dist_on:
            by_comp = os.environ.get("SKDER_AMD_EXCHANGE") == "components"
            out["exchange"] = {"mode": "components: markers all-gathered, seeds to the owner of each connected component" if by_comp else
                               "replicate: raw sketches all-gathered, candidate pairs to the owner of the probed genome"}
            if by_comp:
                out["exchange"].update(getattr(step, "exchange_stats", {}))
            out["exchange"]["other_exchange"] = other_exchange
        out["config"]["us_per_chained_pair"] = 1e3 * (join_ms + step.runs_ms + tm[3] + tm[4] + tm[5]) / max(n_chained, 1.0)
        extras = not args.no_cpu_baseline
        # (rank 0 of an N > 1 job works on alone from here: the process group is gone, the other ranks have left)
        def leg(name, fn):
            """an auxiliary leg never costs the headline line: its failure is recorded under its own key (and on stderr)"""
            try:
                return fn()
            except Exception as ex:
                import traceback
                traceback.print_exc(file=sys.stderr)
                out[name] = {"error": repr(ex)}
                return None
        if extras and args.parity_pairs > 0:
            ps = leg("parity_sample", lambda: parity_sample(recipe, edges, args.parity_pairs))
            if ps is not None:
                out["parity_sample"] = ps
                out["config"]["parity_sample_pairs"] = ps["pairs"]
                out["config"]["parity_sample_mismatches"] = ps["mismatches"]
        if extras:
            gp = leg("parity_vs_skani", lambda: golden_parity(dev))
            if gp is not None:
                out["parity_vs_skani"] = gp
                for k in ("values_equal_at_print_precision", "rows_fully_equal", "tc_listings_identical"):
                    out["config"]["golden_" + k] = gp[k]["equal"]
        batches.clear()      # the headline's resident bases are not needed any more
        torch.cuda.empty_cache()
        if extras:
            import shutil
            # The CPU sample: every second genome of the workload -- with 100 genomes per species (10 strains x 10 isolates) that is
            # every species with 5 strains x 10 isolates, so the share of pairs that pass the screen and are chained is the full
            # workload's (1.96 % against 1.98 %), unlike a block of consecutive genomes (a few whole species: 4.8 x denser)
            sample = list(range(0, N, 2))[:args.cpu_sample_genomes] if args.cpu_sample_genomes > 0 else []
            written = leg("cpu_baseline", lambda: write_workload_sample(engine, ctx, torch, recipe, sample))
            if written is not None:
                tmp, all_paths, sizes = written
                try:
                    if world == 1 and args.e2e_genomes > 0 and len(all_paths) >= 8:
                        leg("end_to_end", lambda: end_to_end_legs(out, tmp, all_paths, sizes, args, dev, total_bases, ms_per_step))
                    if all_paths:
                        cb = leg("cpu_baseline", lambda: cpu_baseline(tmp, all_paths, N, pairs, n_chained_all))
                        if cb is not None:
                            out["cpu_baseline"] = cb
                finally:
                    shutil.rmtree(tmp, ignore_errors=True)
        if world == 1 and not args.no_realistic and extras:
            out["realistic"] = leg("realistic", lambda: realistic_workloads(engine, ctx, torch, synth, args)) or out.get("realistic", {})
            for k, v in out["realistic"].items():            # (scalars under config survive the driver's parse of the line)
                if k.startswith("real_derived_") and isinstance(v, dict) and "us_per_chained_pair" in v:
                    out["config"]["real_derived_us_per_chained_pair"] = v["us_per_chained_pair"]
                    out["config"]["real_derived_genomes"] = v["genomes"]
            if args.low_mem_genomes > 0:
                try:
                    out["realistic"]["low_mem_greedy_%d" % args.low_mem_genomes] = low_mem_greedy_scale(engine, ctx, torch, synth, args.low_mem_genomes, 2_800_000, dev)
                except Exception as ex:
                    out["realistic"]["low_mem_greedy_%d" % args.low_mem_genomes] = {"error": str(ex)}
            if args.one_species_genomes > 0:
                try:
                    torch.cuda.empty_cache()
                    out["realistic"]["low_mem_greedy_one_species"] = low_mem_greedy_one_species(engine, ctx, torch, synth, args.one_species_genomes, dev)
                except Exception as ex:
                    out["realistic"]["low_mem_greedy_one_species"] = {"error": repr(ex)}
        print(json.dumps(out), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
