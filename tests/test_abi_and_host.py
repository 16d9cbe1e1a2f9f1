"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/skder_amd.h declares, host-only logic works, and every compute entry point fails LOUDLY on a
machine without a gfx950 device (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "skder_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(skder_amd_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from skder_amd import _lib
    L = _lib.lib()
    decl = _declared_symbols()
    assert len(decl) >= 20
    for name in decl:
        assert hasattr(L, name), "include/skder_amd.h declares %s but libskder_amd.so does not export it" % name
        assert name in _lib.SYMBOLS, "%s has no ctypes prototype in skder_amd/_lib.py" % name


def test_struct_layouts_match_header():
    from skder_amd import _lib, engine
    assert C.sizeof(_lib.Edge) == 80 and engine.EDGE_DTYPE.itemsize == 80
    assert C.sizeof(_lib.Batch) == 32
    assert C.sizeof(_lib.RawView) == 8 + 3 * 8 + 9 * 8


def test_parse_skani_params():
    """bin/skder:132,199-201: the -p string; '-s X' accepted, every other skani flag rejected loudly"""
    from skder_amd.skder import parse_skani_params
    assert parse_skani_params("-s 89.5") == 89.5
    assert parse_skani_params("") == 80.0
    assert parse_skani_params("  -s   70 ") == 70.0
    for bad in ("--no-learned-ani", "-c 30", "-s", "-s abc", "--robust", "-m 200", "-s 90 --median", "--fast"):
        with pytest.raises(RuntimeError):
            parse_skani_params(bad)


def _no_gpu():
    try:
        import torch
        return not torch.cuda.is_available()
    except Exception:
        return True


@pytest.mark.skipif(not _no_gpu(), reason="checks the behaviour WITHOUT a GPU")
def test_compute_entry_points_fail_loudly_without_gpu(tmp_path):
    """the product path must not fall back to anything when the device is missing; like util.runCmd
    (util.py:636-652) the caller gets a RuntimeError and no output file"""
    import skder_amd
    from skder_amd import engine
    listing = tmp_path / "l.txt"
    listing.write_text("/nonexistent/a.fna\n")
    out = tmp_path / "out.tsv"
    with pytest.raises(RuntimeError, match="skani triangle"):
        skder_amd.runSkaniTriangle(str(listing), str(out), "-s 89.5", 50.0, "greedy", False, None, threads=1)
    assert not out.exists()
    with pytest.raises(RuntimeError, match="no HIP device|CPU fallback"):
        engine.Context(0)
    n50 = tmp_path / "n50.txt"
    n50.write_text("/nonexistent/a.fna\t1000\n")
    with pytest.raises(RuntimeError):
        skder_amd.lowMemGreedyDerep(str(listing), str(tmp_path) + "/", str(n50), str(tmp_path / "res.txt"), str(tmp_path) + "/",
                                    99.0, 90.0, None, threads=1)


def test_batch_layout_alignment():
    from skder_amd import engine
    lay = engine.BatchLayout([np.array([500, 8192, 777], np.uint32), np.array([100000], np.uint32)])
    assert (lay.rec_off % 32 == 0).all() and lay.rec_off[0] == 32
    assert list(lay.genome_rec_begin) == [0, 3, 4]
    assert lay.total_bytes >= int(lay.rec_off[-1]) + 100000 + engine.SKDER_TILE + 32
    with pytest.raises(ValueError):
        engine.BatchLayout([np.array([499], np.uint32)])
    host = lay.pack_host([np.full(500 + 8192 + 777, ord("C"), np.uint8), np.full(100000, ord("G"), np.uint8)])
    assert host[32] == ord("C") and host[int(lay.rec_off[3])] == ord("G") and host[31] == ord("A")


def test_synthetic_recipe_is_deterministic_and_structured():
    from skder_amd import synth
    a, b = synth.make_recipe(40, genome_len=50000, n_species=2), synth.make_recipe(40, genome_len=50000, n_species=2)
    assert np.array_equal(a.lineage, b.lineage) and np.array_equal(a.params, b.params)
    assert all(np.array_equal(x, y) for x, y in zip(a.rec_lens, b.rec_lens))
    assert all((r >= 1000).all() for r in a.rec_lens)
    g0, g1, g20 = (synth.bases_numpy(a, g) for g in (0, 1, 20))
    assert set(np.unique(g0)) <= set(b"ACGT")
    n = min(len(g0), len(g1))
    ident_strain = (g0[:n] == g1[:n]).mean()          # same species, different strain
    ident_isolate = (g0[:n] == synth.bases_numpy(a, 10)[:n]).mean()   # same strain
    m = min(len(g0), len(g20))
    ident_species = (g0[:m] == g20[:m]).mean()        # different species: ~0.25
    assert ident_isolate > ident_strain > 0.6 and ident_species < 0.3
    assert ident_isolate > 0.99


def test_synthetic_recipe_mixed_lengths():
    """len_range (BASELINE config 5's 1-8 Mb shape): species lengths fall in the range, genomes of one species
    share their length, and the default recipe's random stream is untouched by the option"""
    from skder_amd import synth
    r = synth.make_recipe(300, n_species=6, len_range=(100_000, 800_000))
    lens = np.array([r.total_len(g) for g in range(r.n)])
    assert lens.min() >= 100_000 and lens.max() <= 800_000
    for s in range(6):
        assert len(set(lens[r.species == s])) == 1
    assert len(set(lens)) == 6
    a, b = synth.make_recipe(60, genome_len=50000, n_species=3), synth.make_recipe(60, genome_len=50000, n_species=3, len_range=None)
    assert all(np.array_equal(x, y) for x, y in zip(a.rec_lens, b.rec_lens))


def test_sketch_store_staleness_rule(tmp_path):
    """driver._store_is_fresh: a store is reused only while every FASTA file that still exists has the size and
    modification time it had when the store was written; a file that is gone cannot contradict it; other paths or another
    order are another listing"""
    from skder_amd import driver
    files = []
    for i in range(3):
        p = tmp_path / ("g%d.fa" % i)
        p.write_text(">r\n" + "ACGT" * (200 + i) + "\n")
        files.append(str(p))
    then = driver._file_stamps(files)
    assert driver._store_is_fresh(then, driver._file_stamps(files))
    os.utime(files[1], ns=(1, 1))                                    # touched: modification time differs
    assert not driver._store_is_fresh(then, driver._file_stamps(files))
    then = driver._file_stamps(files)
    open(files[2], "a").write("ACGT\n")                              # grown in place
    assert not driver._store_is_fresh(then, driver._file_stamps(files))
    then = driver._file_stamps(files)
    os.remove(files[0])                                              # gone: nothing to compare with
    assert driver._store_is_fresh(then, driver._file_stamps(files))
    assert not driver._store_is_fresh(then, driver._file_stamps(files[::-1]))
    assert not driver._store_is_fresh(then, driver._file_stamps(files[:2]))


def test_bench_gpus_flag_must_agree_with_the_launcher():
    """bench.py --gpus N under a launcher whose WORLD_SIZE differs fails loudly, before torch or the library is imported
    (the flag used to be parsed and never read)"""
    import subprocess
    import sys
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=3" in r.stderr and "--gpus 2" in r.stderr
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "0"], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0


def test_header_lists_exactly_the_switches_the_library_reads():
    """include/skder_amd.h promises ONE list of every environment switch the library reads: the names in that comment equal the
    names the sources pass to getenv (measurement builds behind -DSKDER_CU_MASK_PROBE aside), and the count it states is right"""
    src = os.path.join(ROOT, "skder_amd", "csrc")
    read = set()
    for fn in os.listdir(src):
        if fn.endswith((".hip", ".cpp", ".h")):
            read |= set(re.findall(r'getenv\("(SKDER_AMD_[A-Z0-9_]+)"\)', open(os.path.join(src, fn)).read()))
    read -= {"SKDER_AMD_CU_MASK", "SKDER_AMD_CU_MASK_MODE"}
    head = open(os.path.join(ROOT, "include", "skder_amd.h")).read()
    block = head[head.index("ENVIRONMENT SWITCHES read by the library"):head.index("Read by the Python host mirror")]
    listed = set(re.findall(r"SKDER_AMD_[A-Z0-9_]+", block))
    assert listed == read, (sorted(listed - read), sorted(read - listed))
    words = {13: "thirteen", 14: "fourteen", 15: "fifteen", 16: "sixteen"}
    assert "all %s of them" % words[len(read)] in block
    # and nothing else in the repository's documentation names a switch that no longer exists
    known = read | {"SKDER_AMD_DEVICE", "SKDER_AMD_DEVICES", "SKDER_AMD_SEARCH_BATCH", "SKDER_AMD_SEARCH_ALL", "SKDER_AMD_FORCE_DIST",
                    "SKDER_AMD_DIST_BACKEND", "SKDER_AMD_EXCHANGE", "SKDER_AMD_OTHER_EXCHANGE"}
    named = set(re.findall(r"SKDER_AMD_[A-Z0-9_]+=", open(os.path.join(ROOT, "INTEGRATION.md")).read()))
    assert {n.rstrip("=") for n in named} <= known, sorted({n.rstrip("=") for n in named} - known)


def test_bench_sample_writers_on_stubs(tmp_path):
    """bench.write_workload_sample / write_sample_files (the FASTA files the CPU baseline and the end-to-end legs read) with the device
    replaced by stubs: a `torch` whose tensors are numpy arrays and a context whose synth_fill writes the numpy statement of the
    generator into the device layout.  Every file must hold exactly the genome's records, 80 columns, in record order."""
    import sys
    import types
    sys.path.insert(0, ROOT)
    import bench
    from skder_amd import engine, synth

    class T:
        def __init__(self, n): self.a = np.zeros(n, np.uint8)
        def data_ptr(self): return self
        def cpu(self): return self
        def numpy(self): return self.a
    fake_torch = types.SimpleNamespace(empty=lambda n, dtype=None, device=None: T(n), uint8=None,
                                       cuda=types.SimpleNamespace(synchronize=lambda: None))

    class Ctx:
        def synth_fill(self, t, layout, lineage, params):
            # (lineage / params select the genomes: find them in the recipe by their isolate seed)
            for k in range(layout.n_genomes):
                g = int(np.flatnonzero((rec.lineage == lineage[k]).all(axis=1))[0])
                bases, at = synth.bases_numpy(rec, g), 0
                for r in range(int(layout.genome_rec_begin[k]), int(layout.genome_rec_begin[k + 1])):
                    o, l = int(layout.rec_off[r]), int(layout.rec_len[r])
                    t.a[o:o + l] = bases[at:at + l]
                    at += l
    rec = synth.make_recipe(9, genome_len=30000, n_species=1, strains_per_species=3)
    picked = [0, 2, 4, 6, 8]
    tmp, paths, sizes = bench.write_workload_sample(engine, Ctx(), fake_torch, rec, picked, chunk=2)
    try:
        assert len(paths) == len(picked) and all(os.path.getsize(p) == n for p, n in zip(paths, sizes))
        for g, p in zip(picked, paths):
            assert os.path.basename(p) == "g%05d.fasta" % g
            lens, seq = bench._read_fasta_records(p)
            assert np.array_equal(lens, rec.rec_lens[g]) and np.array_equal(seq, synth.bases_numpy(rec, g))
            body = [l for l in open(p, "rb").read().split(b"\n") if l and not l.startswith(b">")]
            assert max(len(l) for l in body) == 80
    finally:
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
    # the resident-batch form the measurement scripts use
    layout = engine.BatchLayout([rec.rec_lens[g] for g in (1, 3)])
    t = T(layout.total_bytes)
    Ctx().synth_fill(t, layout, rec.lineage[[1, 3]], rec.params[[1, 3]])
    tmp, paths, nbytes = bench.write_sample_files([(layout, t)], 5)
    try:
        assert len(paths) == 2 and nbytes == sum(os.path.getsize(p) for p in paths)
        for g, p in zip((1, 3), paths):
            lens, seq = bench._read_fasta_records(p)
            assert np.array_equal(lens, rec.rec_lens[g]) and np.array_equal(seq, synth.bases_numpy(rec, g))
    finally:
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)


def test_bench_cpu_baseline_leg_on_small_files(oracle, tmp_path, monkeypatch):
    """bench.cpu_baseline (the `cpu_baseline` object of the line) on a dozen small genomes: `value` is the MEASURED rate of one oracle
    triangle over the sample, the extrapolation is a side field, kind is "port" when no skani is installed"""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    from skder_amd import synth
    monkeypatch.setenv("SKANI_REF_NO_SEARCH", "1")
    monkeypatch.setenv("PATH", "/usr/bin:/bin")
    rec = synth.make_recipe(12, genome_len=60000, n_species=2, strains_per_species=3)
    paths = []
    for g in range(0, 12, 2):                      # every second genome, as the bench samples
        p = str(tmp_path / ("g%05d.fasta" % g))
        synth.write_fasta(rec, g, p)
        paths.append(p)
    cb = bench.cpu_baseline(str(tmp_path), paths, 12, 66, 30.0)
    m = cb["measured_on_sample"]
    assert cb["kind"] == "port" and cb["unit"] == "genome-pairs/s" and cb["cores"] >= 1
    assert m["files"] == 6 and m["pairs"] == 15 and m["chained_pairs"] == 6 and m["wall_s"] > 0       # two species x C(3, 2)
    assert cb["value"] == m["pairs_per_s"] == m["pairs"] / m["wall_s"]
    assert cb["extrapolated_full_workload_pairs_per_s"] > 0 and "measured" in cb["value_is"]


def test_bench_cites_only_profile_files_that_exist():
    """every profiles/... file bench.py names in its line or its comments is committed -- except the round's own PMC summary
    (bench.PMC_TRAFFIC), whose absence the line reports as `traffic_source: missing` instead of reading a stale round's file"""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    text = open(os.path.join(ROOT, "bench.py")).read()
    cited = set(re.findall(r"profiles/[A-Za-z0-9_./]+\.(?:json|txt|csv|hip|sh|py)", text))
    missing = sorted(c for c in cited if not os.path.isfile(os.path.join(ROOT, c)) and c != bench.PMC_TRAFFIC.replace(os.sep, "/"))
    assert not missing, missing


def test_profile_summary_tool_on_a_synthetic_collection(tmp_path):
    """profiles/summarise.py on a made-up collection directory (the layout profiles/collect.sh writes): per-kernel counters summed over
    dispatches, calibration factors applied by pattern (the index kernel's FETCH_SIZE through gather16_random), kernels without a pattern
    through the default (streaming reads x the read16 factor), the generator kernel left out of the step total, SQ summaries copied with
    the round's prefix"""
    import csv
    import json
    import shutil
    import subprocess
    import sys
    root = tmp_path / "repo"
    (root / "profiles").mkdir(parents=True)
    shutil.copy(os.path.join(ROOT, "profiles", "summarise.py"), root / "profiles" / "summarise.py")
    src = root / "gpurun_out" / "prof_roundX"

    def counters(sub, rows):
        d = src / sub / "x"
        d.mkdir(parents=True)
        with open(d / "1_counter_collection.csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Kernel_Name", "Counter_Name", "Counter_Value"])
            for name, value in rows:
                w.writerow([name, "X", value])

    def stats(sub):
        d = src / sub / "x"
        d.mkdir(parents=True)
        (d / "1_kernel_stats.csv").write_text("Name,Calls\nk,1\n")
        (d / "1_domain_stats.csv").write_text("Name,Calls\nKERNEL_DISPATCH,1\n")
    stats("stats")
    kb = 1 << 20                                  # counter values are KB
    counters("calib_fetch", [("read16_coalesced(...)", 3 * kb), ("read4_coalesced(...)", 3 * kb), ("gather16_random(...)", 12 * kb)])
    counters("calib_write", [("write4_coalesced(...)", 6 * kb)])
    (src / "calib_fetch.log").write_text("known_bytes read16_coalesced %d\nknown_bytes read4_coalesced %d\nknown_bytes gather16_random %d\n"
                                         "known_bytes write4_coalesced %d\n" % (6 << 30, 6 << 30, 6 << 30, 6 << 30))
    counters("fetch", [("void sketch_tiles_kernel<32781>(unsigned char*)", 100), ("void sketch_tiles_kernel<32781>(unsigned char*)", 50),
                       ("index_genome_lds_kernel(GenomeMeta*)", 40), ("scan_apply_kernel(unsigned int*)", 10), ("synth_fill_kernel(x)", 1000)])
    counters("write", [("void sketch_tiles_kernel<32781>(unsigned char*)", 20), ("index_genome_lds_kernel(GenomeMeta*)", 8),
                       ("scan_apply_kernel(unsigned int*)", 4), ("synth_fill_kernel(x)", 5000)])
    (src / "bench_line.json").write_text(json.dumps({"roofline": {"kernel": "sketch_tiles_kernel", "traffic": None}}) + "\n")
    (src / "pmc_sketch_tiles_kernel.txt").write_text("SQ_WAVES 2 123\n")
    subprocess.run([sys.executable, str(root / "profiles" / "summarise.py"), "roundX"], check=True, capture_output=True, cwd=str(root))
    t = json.load(open(root / "profiles" / "roundX_pmc_traffic.json"))
    sk, ix, sc = t["sketch_tiles_kernel"], t["index_genome_lds_kernel"], t["scan_apply_kernel"]
    assert sk["FETCH_SIZE"]["launches"] == 2 and sk["FETCH_SIZE"]["corrected_bytes"] == 150 * 1024 * 2.0        # read16: the counter shows half
    assert ix["FETCH_SIZE"]["pattern"] == "gather16_random" and ix["FETCH_SIZE"]["corrected_bytes"] == 40 * 1024 * 0.5
    assert "pattern" not in sc["FETCH_SIZE"]
    step = t["__step__"]
    assert step["kernels"] == 3                                                                                    # the generator is not part of a step
    assert step["FETCH_SIZE"] == 150 * 1024 * 2.0 + 40 * 1024 * 0.5 + 10 * 1024 * 2.0 and step["WRITE_SIZE"] == (20 + 8 + 4) * 1024
    assert step["bytes"] == step["FETCH_SIZE"] + step["WRITE_SIZE"]
    assert json.load(open(root / "profiles" / "roundX_bench_line.json"))["roofline"]["traffic"] == 150 * 1024 * 2.0 + 20 * 1024
    assert (root / "profiles" / "roundX_pmc_sketch_tiles_kernel.txt").read_text() == "SQ_WAVES 2 123\n"


def test_bench_reads_its_traffic_figures_from_the_committed_summary():
    """bench.pmc_traffic: the dominant kernel's measured bytes per step and the step total from profiles/round6_pmc_traffic.json; a missing
    or unreadable file gives None and a reason, never an exception"""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    t, src, step = bench.pmc_traffic("sketch_tiles_kernel")
    assert 16.5e9 < t < 17.5e9 and 160e9 < step < 175e9 and "round 5" in src.lower() or "ROUND 5" in src     # 1.07 x the 15.96 GB the algorithm moves
    assert step > t
    assert bench.pmc_traffic("sketch_tiles_kernel", "profiles/no_such_file.json")[::2] == (None, None)
    assert bench.pmc_traffic("no_such_kernel")[0] is None
