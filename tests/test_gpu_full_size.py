"""BASELINE.json configs[2], [3], [4] AT THEIR OWN SIZES on one MI355X, asserted.

The oracle cannot cross-check 12.5 M / 200 M / 1.25 G pairs one by one; each test therefore holds
  * counts and the structure of the pair set that the generator dictates (every within-species pair, nothing across species),
  * size-independent properties of the path (bit-identical reruns, the --min-af rule on unrounded values, the greedy listing's
    invariants, speculative == sequential search loop),
  * SAMPLED pairs bit-equal with the oracle (bases regenerated on the host by the numpy statement of the generator),
  * a committed DIGEST of the representative listing (tests/golden/full_size_digests.json, written on the GPU box by
    `SKDER_AMD_WRITE_DIGESTS=<file> pytest ...`): integer generator + integer engine, so the listing is reproducible bit for bit.
A missing digest FAILS the test (never a silent pass).  Together the three tests take about two minutes on the GPU box."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

DIGESTS = os.path.join(GOLDEN, "full_size_digests.json")


@pytest.fixture()
def gpu():
    """a context of its own per test, and the device memory the library and torch cached handed back before and after: the
    50,000-genome test needs two thirds of the HBM"""
    import torch
    assert torch.cuda.is_available(), "no GPU visible"
    from skder_amd import engine, _lib
    torch.cuda.empty_cache()
    _lib.lib().skder_amd_release_cached_buffers(0)
    ctx = engine.Context(0)
    yield engine, ctx, torch
    ctx.close()
    torch.cuda.empty_cache()
    _lib.lib().skder_amd_release_cached_buffers(0)


def _digest_check(key, listing):
    """sha256 of the listing (one index per line) against the committed figure; SKDER_AMD_WRITE_DIGESTS=<file> records it instead"""
    got = hashlib.sha256("".join("%d\n" % int(r) for r in listing).encode()).hexdigest()
    dump = os.environ.get("SKDER_AMD_WRITE_DIGESTS")
    if dump:
        have = json.load(open(dump)) if os.path.isfile(dump) else {}
        have[key] = {"sha256": got, "representatives": len(listing)}
        os.makedirs(os.path.dirname(os.path.abspath(dump)), exist_ok=True)
        json.dump(have, open(dump, "w"), indent=1, sort_keys=True)
        return
    assert os.path.isfile(DIGESTS), "tests/golden/full_size_digests.json is missing: nothing to hold the listing against"
    want = json.load(open(DIGESTS))
    assert key in want, "no committed digest for %s" % key
    assert want[key]["representatives"] == len(listing) and want[key]["sha256"] == got, (key, len(listing), got)


def _streamed_sketches(gpu, rec, step):
    """the recipe's genomes generated on the device, sketched and dropped batch by batch (the bases are never resident together)"""
    engine, ctx, torch = gpu
    total = sum(rec.total_len(g) for g in range(rec.n))
    sk = engine.Sketches(ctx)
    sk.reserve(total // 120, total // 900)
    for b0 in range(0, rec.n, step):
        gs = range(b0, min(b0 + step, rec.n))
        layout = engine.BatchLayout([rec.rec_lens[g] for g in gs])
        d = torch.empty(layout.total_bytes, dtype=torch.uint8, device="cuda")
        ctx.synth_fill(d.data_ptr(), layout, rec.lineage[gs.start:gs.stop], rec.params[gs.start:gs.stop])
        sk.sketch_batch(d.data_ptr(), layout)
        del d
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    return sk


def _n50(lens):
    ls = sorted((int(x) for x in lens), reverse=True)
    half, cum = int(sum(ls) / 2), 0
    for l in ls:
        cum += l
        if cum >= half:
            return l
    return 0


def _sampled_pairs_equal_oracle(oracle, synth, rec, rows, trios):
    """every pair inside each trio of genomes: the row's integer sums and doubles against oracle.pair, bit for bit"""
    p = oracle.default_params()
    got = {(int(e["ref"]), int(e["query"])): e for e in rows if int(e["ref"]) in trios["all"] and int(e["query"]) in trios["all"]}
    n = 0
    for trio in trios["sets"]:
        og = [oracle.Genome.from_bases(synth.bases_numpy(rec, g), rec.rec_lens[g], p) for g in trio]
        for i in range(len(trio)):
            for j in range(i + 1, len(trio)):
                a, b = trio[i], trio[j]
                assert a < b
                r = oracle.pair(og[i], og[j], p)
                e = got[(a, b)]
                assert int(e["sum_anchors"]) == r.sum_anchors and int(e["sum_seeds"]) == r.sum_seeds and int(e["cell_seeds"]) == r.cell_seeds, (a, b)
                assert int(e["aligned_bases"]) == r.aligned_bases and int(e["n_chains"]) == r.n_chains, (a, b)
                assert float(e["ani"]) == r.ani and float(e["ani_raw"]) == r.ani_raw, (a, b)
                assert float(e["af_ref"]) == r.af_ref and float(e["af_query"]) == r.af_query, (a, b)
                n += 1
    return n


def test_config3_dynamic_on_5000_genomes(gpu, oracle):
    """configs[2]: 5,000 synthetic 3 Mb genomes (50 species x 10 strains x 10 isolates), resident sketches -> skder_amd_db_triangle ->
    skder_amd_select_dynamic (skDERcore.cpp).  247,500 rows = every within-species pair; 21 sampled pairs bit-equal with the oracle;
    dynamic and greedy listings: every species represented, no genome twice, greedy never keeps two isolates of one strain at
    -i 98.5 (within a strain the model ANI is >= 98.7, across strains <= 97.6); the dynamic listing's digest is committed."""
    from skder_amd import synth, selection as S
    from skder_amd.skder import Database
    n, nsp = 5000, 50
    rec = synth.make_recipe(n, genome_len=3_000_000)
    sk = _streamed_sketches(gpu, rec, 1250)
    paths = ["/synthetic/species%03d/g%05d.fasta" % (int(rec.species[g]), g) for g in range(n)]
    n50 = [_n50(rec.rec_lens[g]) for g in range(n)]
    with Database.from_sketches(sk, paths, n50) as db:
        sk.close()
        rows = db.triangle(50.0, 80.0)
        again = db.triangle(50.0, 80.0)
    assert np.array_equal(rows, again)                                                   # bit-identical reruns, row order included
    per = n // nsp
    assert len(rows) == nsp * (per * (per - 1) // 2) == 247_500
    assert (rows["ref"] < rows["query"]).all() and (rows["ref"] // per == rows["query"] // per).all()
    assert len({(int(a), int(b)) for a, b in zip(rows["ref"], rows["query"])}) == len(rows)
    strain = lambda g: (g % per) % 10
    same = (rows["ref"] % per) % 10 == (rows["query"] % per) % 10
    assert rows["ani"][same].min() > rows["ani"][~same].max() and rows["ani"][same].min() > 0.98
    # sampled pairs: in seven species two isolates of one strain and one of another
    sets = [[sp * per + 3, sp * per + 13, sp * per + 24] for sp in (0, 7, 19, 23, 31, 42, 49)]
    checked = _sampled_pairs_equal_oracle(oracle, synth, rec, rows, {"sets": sets, "all": {g for t in sets for g in t}})
    assert checked == 21
    dyn = S.native_dynamic(rows, paths, n50, 99.5, 50.0, 10.0)
    assert len(set(dyn)) == len(dyn) and {r // per for r in dyn} == set(range(nsp))
    gre = S.native_greedy(rows, paths, n50, 98.5, 50.0)
    assert {r // per for r in gre} == set(range(nsp)) and len({(r // per, strain(r)) for r in gre}) == len(gre) == nsp * 10
    _digest_check("config3_dynamic_99.5_50_10", dyn)
    _digest_check("config3_greedy_98.5_50", gre)


def test_config4_low_mem_greedy_on_20000_genomes(gpu, tmp_path, monkeypatch):
    """configs[3]: lowMemGreedyDerep -i 99.5 -f 50 (skder.py:95-134) over 20,000 synthetic 2.8 Mb genomes (200 species), the sketch
    database resident.  The speculative batches over the live set, the speculative batches with every row computed and the
    reference's loop call for call (one search + TSV + parse per representative) give ONE listing; every species is represented;
    and on a sample of representatives the greedy invariant holds on the rows themselves: no representative chosen later lies
    within the cut-offs of one chosen earlier (it would have been accounted for)."""
    import ctypes as C
    import skder_amd
    from skder_amd import synth, _lib
    from skder_amd.skder import Database
    n, nsp = 20000, 200
    rec = synth.make_recipe(n, genome_len=2_800_000)
    sk = _streamed_sketches(gpu, rec, 1250)
    paths = ["/synthetic/species%03d/g%05d.fasta" % (int(rec.species[g]), g) for g in range(n)]
    n50 = [_n50(rec.rec_lens[g]) for g in range(n)]
    listing, n50_file = tmp_path / "listing.txt", tmp_path / "Concatenated_N50.txt"
    listing.write_text("".join(p + "\n" for p in paths))
    n50_file.write_text("".join("%s\t%d\n" % kv for kv in zip(paths, n50)))
    with Database.from_sketches(sk, paths, n50) as db:
        sk.close()
        out = {}
        for tag, width, env in (("live", 0, None), ("all_rows", 0, "1"), ("sequential", 1, None)):
            if env:
                monkeypatch.setenv("SKDER_AMD_SEARCH_ALL", env)
            ws = tmp_path / tag
            ws.mkdir()
            res = ws / "reps.txt"
            skder_amd.lowMemGreedyDerep(str(listing), str(ws) + "/", str(n50_file), str(res), str(ws) + "/", 99.5, 50.0, None,
                                        search_batch=width, database=db)
            if env:
                monkeypatch.delenv("SKDER_AMD_SEARCH_ALL")
            out[tag] = res.read_text()
        assert out["live"] == out["all_rows"] == out["sequential"]
        reps = out["live"].split()
        index_of = {p: i for i, p in enumerate(paths)}
        idx = [index_of[r] for r in reps]
        per = n // nsp
        assert len(set(idx)) == len(idx) and {i // per for i in idx} == set(range(nsp))
        # the order the loop handled the genomes in: N50 descending, stable (skder.py:116)
        order = sorted(range(n), key=lambda g: -float(n50[g]))
        turn = np.empty(n, np.int64)
        turn[order] = np.arange(n)
        is_rep = np.zeros(n, bool)
        is_rep[idx] = True
        rng = np.random.RandomState(4)
        sample = [reps[i] for i in rng.choice(len(reps), 48, replace=False)]
        rows = db.search_batch(sample)
        ok = np.zeros(len(rows), np.uint8)
        _lib.lib().skder_amd_rows_pass(rows.ctypes.data, len(rows), 99.5, 50.0, 5, ok.ctypes.data)
        assert len(rows) > 48 * 50
        for e, good in zip(rows, ok):
            q, r = index_of[sample[int(e["query"])]], int(e["ref"])
            if good and r != q and is_rep[r]:
                assert turn[r] < turn[q], (paths[q], paths[r])        # a representative within the cut-offs of q was chosen BEFORE q, never after
        # a search lists the query itself (skder.py:127-129 marks it through its own row)
        self_rows = [(index_of[sample[int(e["query"])]], int(e["ref"])) for e in rows]
        assert all((index_of[s], index_of[s]) in set(self_rows) for s in sample)
    _digest_check("config4_low_mem_greedy_99.5_50", idx)


def test_config5_mixed_sizes_on_50000_genomes(gpu, oracle, tmp_path):
    """configs[4]: 50,000 synthetic genomes of 1 - 8 Mb in 500 species (223 Gb of bases streamed through the sketch stage), the
    aligned-fraction filter on, one GPU.  The unfiltered triangle holds exactly the 2,475,000 within-species pairs; the filtered call
    keeps exactly the rows whose max(AF_ref, AF_query) reaches the cut-off on the unrounded single-precision values, in
    the unfiltered order, and its text table has those rows; same-strain pairs are closer than different strains; sampled pairs of
    the longest and the shortest species equal the oracle's bit for bit; the greedy listing's digest is committed."""
    from skder_amd import synth, selection as S
    from skder_amd.skder import Database
    engine, ctx, torch = gpu
    n, nsp = 50000, 500
    free, total = torch.cuda.mem_get_info()
    # a FAILURE, not a skip: config 5 is one of BASELINE.json's five configs and an MI355X holds it (288 GB); a box that cannot is not the
    # hardware this repository is measured on, and a skip would read as green
    assert free >= 215e9, "BASELINE config 5 needs 215 GB of free HBM; %.0f of %.0f GB are free on this device" % (free / 1e9, total / 1e9)
    rec = synth.make_recipe(n, len_range=(1_000_000, 8_000_000))
    sk = _streamed_sketches(gpu, rec, 1000)
    paths = ["/mixed/species%03d/g%05d.fasta" % (int(rec.species[g]), g) for g in range(n)]
    n50 = [_n50(rec.rec_lens[g]) for g in range(n)]
    per = n // nsp
    with Database.from_sketches(sk, paths, n50) as db:
        sk.close()
        e0 = db.triangle(0.0, 80.0)
        assert len(e0) == nsp * (per * (per - 1) // 2) == 2_475_000
        assert (e0["ref"] < e0["query"]).all() and (e0["ref"] // per == e0["query"] // per).all()
        mx = np.maximum(e0["af_ref"].astype(np.float32), e0["af_query"].astype(np.float32)).astype(np.float64) * 100.0
        cut = float(np.median(mx))
        out = tmp_path / "filtered.tsv"
        ef = db.triangle(cut, 80.0, out_tsv=str(out))
        keep = mx >= cut
        assert 0.3 < keep.mean() < 0.7 and np.array_equal(ef, e0[keep])
        with open(out) as f:
            assert sum(1 for _ in f) == 1 + int(keep.sum())
    same = (e0["ref"] % per) % 10 == (e0["query"] % per) % 10
    assert e0["ani"][same].min() > e0["ani"][~same].max() and e0["ani"][same].min() > 0.98
    assert (e0["af_ref"] > 0).all() and (e0["af_ref"] <= 1).all() and (e0["af_query"] > 0).all() and (e0["af_query"] <= 1).all()
    tl = np.array([rec.total_len(g) for g in range(n)])
    assert tl.max() > 7_500_000 and tl.min() < 1_200_000
    sets = []
    for g in (int(np.argmax(tl)), int(np.argmin(tl)), n // 2):
        g0 = (g // per) * per
        sets.append([g0, g0 + 1, g0 + 10])
    checked = _sampled_pairs_equal_oracle(oracle, synth, rec, e0, {"sets": sets, "all": {g for t in sets for g in t}})
    assert checked == 9
    reps = S.native_greedy(e0, paths, n50, 98.5, 50.0)
    strain = lambda g: (g % per) % 10
    assert {r // per for r in reps} == set(range(nsp)) and len({(r // per, strain(r)) for r in reps}) == len(reps)
    _digest_check("config5_greedy_98.5_50", reps)
