"""CPU tests of the NATIVE selection (skder_amd/csrc/select.cpp behind skder_amd_select_greedy / _dynamic / _clusters):
host code, no GPU.  It must be byte-identical with the reference's golden files, with skder_amd/selection.py (the readable statement of
the same rules), with the reference's own binaries where oracle/_ref is built, and with the MGE-mapped branches of the reference's Python
(tests/golden/make_generated.py ran the imported reference with a name mapping); its rounding must be printf's."""
import ctypes as C
import os
import subprocess
import time

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

D = os.path.join(GOLDEN, "downstream")
ANI_CUTS = [90.0, 95.0, 97.0, 98.0, 99.0, 99.5]
AF_CUTS = [10.0, 25.0, 50.0, 75.0, 90.0]
TABLES = {"G1": ("G1_triangle_minaf50_s89.tsv", "skder_results__Concatenated_N50.txt"),
          "G5": ("G5_triangle_minaf10_s89.5.tsv", "skder_gtdb_results__Concatenated_N50.txt")}


def _lines(path):
    with open(path) as f:
        return [l.rstrip("\n") for l in f]


def _load(tag):
    from skder_amd import selection as S
    n50 = S.read_n50(os.path.join(D, TABLES[tag][1]))
    paths = list(n50)
    rows = S.rows_from_table(os.path.join(GOLDEN, TABLES[tag][0]), paths)
    return S, rows, paths, [n50[p] for p in paths], S.edges_from_table(os.path.join(GOLDEN, TABLES[tag][0])), n50


def test_rounding_is_printf(tmp_path):
    """hundredths of a percent of (float)fraction * 100.0f exactly as `%.2f` prints them: random fractions, every float next to a
    tie (x.xx5 boundaries), the ends of the range"""
    from skder_amd import _lib
    f = _lib.lib().skder_amd_pct2_cents
    rng = np.random.RandomState(1)
    vals = [np.float32(x) for x in rng.rand(200000)]
    for k in range(0, 10001, 7):          # the floats around every k + 0.5 hundredths
        t = np.float32((k + 0.5) / 10000.0)
        for d in range(-3, 4):
            v = t
            for _ in range(abs(d)):
                v = np.nextafter(v, np.float32(2.0 if d > 0 else -1.0))
            vals.append(np.float32(v))
    vals += [np.float32(0.0), np.float32(1.0), np.float32(1e-7), np.float32(0.99995), np.float32(0.999949), np.float32(0.5), np.float32(0.125)]
    for v in vals:
        if v < 0:
            continue
        want = int(round(float("%.2f" % float(np.float32(v) * np.float32(100))) * 100))
        assert f(C.c_float(float(v))) == want, float(v)


def test_native_greedy_reproduces_the_reference_run(tmp_path):
    """`skder -g ... -n -i 99.0` from golden G1: the four files of the greedy flow and the clustering table, byte for byte"""
    S, rows, paths, n50, _, _ = _load("G1")
    info, srt, res, clu = (str(tmp_path / n) for n in ("info.txt", "sorted.txt", "res.txt", "clu.txt"))
    reps = S.native_greedy(rows, paths, n50, 99.0, 50.0, info, srt, res)
    assert _lines(info) == _lines(os.path.join(D, "skder_results__Genome_Information_for_Greedy_Clustering.txt"))
    assert _lines(srt) == _lines(os.path.join(D, "skder_results__Genome_Information_for_Greedy_Clustering.sorted.txt"))
    assert _lines(res) == _lines(os.path.join(D, "skder_results__skDER_Results.txt"))
    assert [paths[r] for r in reps] == _lines(res)
    S.native_clusters(rows, paths, reps, 50.0, 99.0, clu)
    assert _lines(clu) == _lines(os.path.join(D, "skder_results__skDER_Clustering.txt"))


def test_native_greedy_reproduces_all_30_cutoff_files():
    S, rows, paths, n50, _, _ = _load("G5")
    for a in ANI_CUTS:
        for f in AF_CUTS:
            want = _lines(os.path.join(D, "tc", "skDER_Results_ANI%s_AF%s.txt" % (a, f)))
            assert [paths[r] for r in S.native_greedy(rows, paths, n50, a, f)] == want, (a, f)


def test_native_dynamic_and_clusters_reproduce_the_generated_goldens(tmp_path):
    """skDERcore's listings and determineClusters' tables as the reference itself produced them (tests/golden/make_generated.py)"""
    from skder_amd import selection as S
    d = os.path.join(D, "generated")
    loaded = {t: _load(t) for t in TABLES}
    seen = 0
    for fn in sorted(os.listdir(d)):
        if fn.startswith("dynamic__"):
            _, tag, rest = fn[:-4].split("__")
            ani, af, maxd = (float(x[len(k):]) for x, k in zip(rest.split("_"), ("ANI", "AF", "D")))
            _, rows, paths, n50, _, _ = loaded[tag]
            out = str(tmp_path / "dyn.txt")
            reps = S.native_dynamic(rows, paths, n50, ani, af, maxd, out)
            assert _lines(out) == _lines(os.path.join(d, fn)), fn
            assert [paths[r] for r in reps] == _lines(out)
            seen += 1
        elif fn.startswith("clusters__"):
            _, tag, mode, rest = fn[:-4].split("__")
            ani, af = (float(x[len(k):]) for x, k in zip(rest.split("_"), ("ANI", "AF")))
            _, rows, paths, n50, _, _ = loaded[tag]
            idx = {p: i for i, p in enumerate(paths)}
            reps = [idx[r] for r in _lines(os.path.join(d, "reps__" + fn[len("clusters__"):]))]
            out = str(tmp_path / "clu.txt")
            S.native_clusters(rows, paths, reps, af, ani, out)
            assert _lines(out) == _lines(os.path.join(d, fn)), fn
            seen += 1
    assert seen == 22


def test_native_equals_the_python_statement_on_random_tables(tmp_path):
    """random tables with ties everywhere (values on a coarse grid, equal N50s, cut-offs on the grid): every output of the native
    selection equals skder_amd/selection.py's, including the mapped names of the MGE branches"""
    from collections import OrderedDict
    from skder_amd import selection as S
    from skder_amd.engine import EDGE_DTYPE
    rng = np.random.RandomState(7)
    for trial in range(12):
        n = int(rng.randint(5, 60))
        paths = ["/data/g%03d%s.fna" % (i, "x" * int(rng.randint(0, 3))) for i in range(n)]
        shown = ["/unprocessed/" + os.path.basename(p) for p in paths]
        n50 = [int(rng.choice([1000, 5000, 5000, 123456, 4350491, 4350494])) for _ in range(n)]
        pairs = [(i, j) for i in range(n) for j in range(i + 1, n) if rng.rand() < 0.5]
        rng.shuffle(pairs)
        rows = np.zeros(len(pairs), EDGE_DTYPE)
        grid = np.array([0.9, 0.95, 0.9712, 0.98, 0.99, 0.99499, 0.995, 0.99501, 1.0], np.float32)
        for k, (i, j) in enumerate(pairs):
            rows[k]["ref"], rows[k]["query"] = i, j
            rows[k]["ani"] = float(rng.choice(grid))
            rows[k]["af_ref"] = float(rng.choice([0.1, 0.25, 0.5, 0.50001, 0.75, 0.9, 0.95, 1.0]))
            rows[k]["af_query"] = float(rng.choice([0.1, 0.25, 0.5, 0.49999, 0.75, 0.9, 0.95, 1.0]))
        edges = S.edges_from_engine(rows, paths)
        nd = OrderedDict(zip(paths, n50))
        for ani, af, maxd in ((99.5, 50.0, 10.0), (99.0, 90.0, 0.0), (95.0, 25.0, 5.0)):
            info, srt, res = (str(tmp_path / x) for x in ("i.txt", "s.txt", "r.txt"))
            reps = S.native_greedy(rows, paths, n50, ani, af, info, srt, res, display=shown)
            want_info = S.genome_information(edges, nd, ani, af)
            assert _lines(info) == want_info
            assert _lines(srt) == S.sort_like_coreutils(want_info)
            want_reps = S.greedy(S.sort_like_coreutils(want_info))
            assert [paths[r] for r in reps] == want_reps
            assert _lines(res) == ["/unprocessed/" + os.path.basename(p) for p in want_reps]      # skder.py:160-163
            dyn = S.native_dynamic(rows, paths, n50, ani, af, maxd, res, display=shown)
            assert [paths[r] for r in dyn] == S.dynamic(edges, nd, ani, af, maxd)
            for rr in (reps, dyn):
                clu = str(tmp_path / "c.txt")
                S.native_clusters(rows, paths, rr, af, ani, clu)
                assert _lines(clu) == S.determine_clusters([paths[r] for r in rr], edges, af, ani)
                S.native_clusters(rows, paths, rr, af, ani, clu, display=shown)                      # skder.py:236-253
                want = S.determine_clusters([paths[r] for r in rr], edges, af, ani)
                mapped = [want[0]]
                for l in want[1:]:
                    c = l.split("\t")
                    c[0] = "/unprocessed/" + os.path.basename(c[0])
                    c[1] = ", ".join("/unprocessed/" + os.path.basename(x) for x in c[1].split(", "))
                    mapped.append("\t".join(c))
                assert _lines(clu) == mapped


def test_native_against_the_reference_binaries(ref_bins, tmp_path):
    """skDERsum / skDERcore compiled from the reference's sources, on the golden tables: identical output"""
    if not ref_bins:
        pytest.skip("the reference's binaries need /root/reference (absent on this box); the golden files they wrote are tested above")
    for tag in TABLES:
        S, rows, paths, n50, _, _ = _load(tag)
        tp, nf = os.path.join(GOLDEN, TABLES[tag][0]), os.path.join(D, TABLES[tag][1])
        for a, f in ((99.0, 50.0), (99.5, 90.0), (97.0, 25.0)):
            out = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "skDERsum"), tp, nf, str(a), str(f)], capture_output=True, text=True, check=True).stdout.splitlines()
            info = str(tmp_path / "i.txt")
            S.native_greedy(rows, paths, n50, a, f, info)
            assert _lines(info) == out
            for maxd in (10.0, 0.0):
                out = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "skDERcore"), tp, nf, str(a), str(f), str(maxd)], capture_output=True, text=True,
                                     check=True).stdout.splitlines()
                assert [paths[r] for r in S.native_dynamic(rows, paths, n50, a, f, maxd)] == out


def test_native_selection_rate():
    """2 * 10^6 synthetic rows over 20,000 genomes: the in-memory selection (no text files) runs at tens of millions of rows per second"""
    from skder_amd import selection as S
    from skder_amd.engine import EDGE_DTYPE
    rng = np.random.RandomState(3)
    n, m = 20000, 2000000
    rows = np.zeros(m, EDGE_DTYPE)
    a = rng.randint(0, n, m).astype(np.uint32)
    b = rng.randint(0, n, m).astype(np.uint32)
    b = np.where(a == b, (b + 1) % n, b).astype(np.uint32)
    rows["ref"], rows["query"] = np.minimum(a, b), np.maximum(a, b)
    rows["ani"] = 0.95 + 0.05 * rng.rand(m)
    rows["af_ref"] = 0.4 + 0.6 * rng.rand(m)
    rows["af_query"] = 0.4 + 0.6 * rng.rand(m)
    paths = ["/genomes/g%06d.fna" % i for i in range(n)]
    n50 = rng.randint(10000, 3000000, n)
    t0 = time.perf_counter()
    reps = S.native_greedy(rows, paths, n50, 99.0, 50.0)
    t1 = time.perf_counter()
    dyn = S.native_dynamic(rows, paths, n50, 99.0, 50.0, 10.0)
    t2 = time.perf_counter()
    assert 0 < len(reps) < n and 0 < len(dyn) < n
    rate_g, rate_d = m / (t1 - t0), m / (t2 - t1)
    print("native greedy %.1f M rows/s, dynamic %.1f M rows/s" % (rate_g / 1e6, rate_d / 1e6))
    assert rate_g > 3e6 and rate_d > 3e6        # (a loaded CI core; measured 30-60 M rows/s)
