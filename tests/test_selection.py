"""CPU tests of the selection counterpart: skder_amd/selection.py must reproduce the
reference's down-stream outputs from the reference's own golden edge tables byte for byte, agree with
the reference's compiled helpers (oracle/_ref, built from /root/reference/src/skDER/*.cpp by
oracle/build_ref.sh) where those are present, and -- fed with the ORACLE's edge table instead of
skani's -- give the same representative listings wherever the goldens are not knife-edge."""
import os
import subprocess

import pytest

from conftest import GOLDEN, ROOT, load_table

D = os.path.join(GOLDEN, "downstream")
ANI_CUTS = [90.0, 95.0, 97.0, 98.0, 99.0, 99.5]      # bin/skder:54-55
AF_CUTS = [10.0, 25.0, 50.0, 75.0, 90.0]


def _lines(path):
    with open(path) as f:
        return [l.rstrip("\n") for l in f]


def test_greedy_chain_reproduces_skder_results_run():
    """`skder -g ... -n -i 99.0` (greedy, -f 50): every intermediate and final file from golden G1"""
    from skder_amd import selection as S
    edges = S.edges_from_table(os.path.join(GOLDEN, "G1_triangle_minaf50_s89.tsv"))
    n50 = S.read_n50(os.path.join(D, "skder_results__Concatenated_N50.txt"))
    info = S.genome_information(edges, n50, 99.0, 50.0)
    assert info == _lines(os.path.join(D, "skder_results__Genome_Information_for_Greedy_Clustering.txt"))
    srt = S.sort_like_coreutils(info)
    assert srt == _lines(os.path.join(D, "skder_results__Genome_Information_for_Greedy_Clustering.sorted.txt"))
    reps = S.greedy(srt)
    assert reps == _lines(os.path.join(D, "skder_results__skDER_Results.txt"))
    clus = S.determine_clusters(reps, edges, 50.0, 99.0)
    assert clus == _lines(os.path.join(D, "skder_results__skDER_Clustering.txt"))


def test_greedy_reproduces_all_30_cutoff_files():
    """the -tc sweep of the GTDB run: 6 ANI x 5 AF cut-offs from golden G5"""
    from skder_amd import selection as S
    edges = S.edges_from_table(os.path.join(GOLDEN, "G5_triangle_minaf10_s89.5.tsv"))
    n50 = S.read_n50(os.path.join(D, "skder_gtdb_results__Concatenated_N50.txt"))
    for a in ANI_CUTS:
        for f in AF_CUTS:
            want = _lines(os.path.join(D, "tc", "skDER_Results_ANI%s_AF%s.txt" % (a, f)))
            assert S.greedy_from_edges(edges, n50, a, f) == want, (a, f)


def _generated():
    d = os.path.join(D, "generated")
    return d, sorted(os.listdir(d))


def test_dynamic_mode_reproduces_the_generated_goldens():
    """skDERcore's listings (tests/golden/make_generated.py: the reference's own binary on G1 / G5) from selection.dynamic,
    byte for byte -- the reference's test run holds no dynamic-mode output of its own"""
    from skder_amd import selection as S
    d, names = _generated()
    tables = {"G1": ("G1_triangle_minaf50_s89.tsv", "skder_results__Concatenated_N50.txt"),
              "G5": ("G5_triangle_minaf10_s89.5.tsv", "skder_gtdb_results__Concatenated_N50.txt")}
    seen = 0
    for fn in names:
        if not fn.startswith("dynamic__"):
            continue
        _, tag, rest = fn[:-4].split("__")
        ani, af, maxd = (float(x[len(k):]) for x, k in zip(rest.split("_"), ("ANI", "AF", "D")))
        edges, n50 = S.edges_from_table(os.path.join(GOLDEN, tables[tag][0])), S.read_n50(os.path.join(D, tables[tag][1]))
        assert S.dynamic(edges, n50, ani, af, maxd) == _lines(os.path.join(d, fn)), fn
        seen += 1
    assert seen == 12


def test_clustering_reproduces_the_generated_goldens():
    """skDER_Clustering.txt as the imported reference's determineClusters writes it for the greedy and the dynamic
    listings of G1 / G5 (tests/golden/make_generated.py)"""
    from skder_amd import selection as S
    d, names = _generated()
    tables = {"G1": "G1_triangle_minaf50_s89.tsv", "G5": "G5_triangle_minaf10_s89.5.tsv"}
    seen = 0
    for fn in names:
        if not fn.startswith("clusters__"):
            continue
        _, tag, mode, rest = fn[:-4].split("__")
        ani, af = (float(x[len(k):]) for x, k in zip(rest.split("_"), ("ANI", "AF")))
        reps = _lines(os.path.join(d, "reps__" + fn[len("clusters__"):]))
        edges = S.edges_from_table(os.path.join(GOLDEN, tables[tag]))
        assert S.determine_clusters(reps, edges, af, ani) == _lines(os.path.join(d, fn)), fn
        seen += 1
    assert seen == 10


def test_against_reference_binaries(ref_bins, tmp_path):
    """skDERsum / skDERcore compiled from the reference's sources: identical stdout"""
    from skder_amd import selection as S
    if not ref_bins:
        pytest.skip("the reference's binaries need /root/reference (absent on this box); the golden files they wrote are tested above")
    for table, n50f in (("G1_triangle_minaf50_s89.tsv", "skder_results__Concatenated_N50.txt"),
                        ("G5_triangle_minaf10_s89.5.tsv", "skder_gtdb_results__Concatenated_N50.txt")):
        tp, nf = os.path.join(GOLDEN, table), os.path.join(D, n50f)
        edges, n50 = S.edges_from_table(tp), S.read_n50(nf)
        for a, f in ((99.0, 50.0), (99.5, 90.0), (97.0, 25.0)):
            out = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "skDERsum"), tp, nf, str(a), str(f)],
                                 capture_output=True, text=True, check=True).stdout.splitlines()
            assert S.genome_information(edges, n50, a, f) == out
            for maxd in (10.0, 0.0):
                out = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "skDERcore"), tp, nf, str(a), str(f), str(maxd)],
                                     capture_output=True, text=True, check=True).stdout.splitlines()
                assert S.dynamic(edges, n50, a, f, maxd) == out, (table, a, f, maxd)


def test_oracle_table_gives_the_reference_listings_where_not_knife_edge(oracle, tmp_path):
    """end to end on the 34 genomes: oracle edge table -> greedy listings vs the 30 golden files.
    ANI/AF differ from skani's in the second decimal, so cut-offs that fall
    inside the residual of a deciding edge can flip; the test counts identical listings and requires
    the robust majority, and set-similarity for the rest."""
    from skder_amd import selection as S
    listing = tmp_path / "l.txt"
    names = sorted(os.listdir(os.path.join(GOLDEN, "genomes")))
    listing.write_text("".join(os.path.join(GOLDEN, "genomes", n) + "\n" for n in names))
    out = tmp_path / "tri.tsv"
    oracle.triangle(str(listing), 10.0, 89.5, 8, str(out), oracle.default_params())
    edges = [(os.path.basename(a), os.path.basename(b), x, y, z) for a, b, x, y, z in S.edges_from_table(str(out))]
    n50 = S.read_n50(os.path.join(D, "skder_gtdb_results__Concatenated_N50.txt"))
    same = 0
    for a in ANI_CUTS:
        for f in AF_CUTS:
            want = _lines(os.path.join(D, "tc", "skDER_Results_ANI%s_AF%s.txt" % (a, f)))
            got = S.greedy_from_edges(edges, n50, a, f)
            same += got == want
            # the NUMBER of representatives is stable everywhere (the species' structure is recovered) ...
            assert abs(len(got) - len(want)) <= 1, (a, f, len(got), len(want))
            # ... and at the cut-offs skDER is actually run with (default -i 99.5; its own test -i 99.0) the
            # listing is identical, order included
            if a >= 99.0:
                assert got == want, (a, f)
    # 97-98 % cut-offs sit inside the bulk of this species' pair ANIs, where the 0.16-point residual of the
    # restatement flips individual edges: measured 19 of 30 listings identical
    assert same >= 18, "only %d of 30 listings identical" % same


def test_listing_at_99_hangs_on_the_model_constants(oracle):
    """How much the golden representative listings depend on the two constants of the ANI stand-in
    (include/skder_amd_spec.h ANI_CAL_CELL / ANI_CAL_SPAN).  The 561 pairs' integer-derived divergences come from the
    oracle once; the model line is then applied for every (cell, span) within +-0.01 of the least-squares optimum
    and the 30 `-tc` listings are recomputed.  What this pins, and what it admits:
      * -i 99.5 (skDER's default): identical for EVERY constant pair of the grid -- robust;
      * -i 99.0 (the reference's own test run): identical only on one side of a line that passes through the optimum --
        the deciding edge (skani 99.13) lies 0.13 points from the cut-off, inside the stand-in's residual.  The shipped
        constants are on the reproducing side; the listing equality at 99.0 is therefore a fit, not evidence of margin."""
    import importlib.util
    import numpy as np
    from skder_amd import selection as S
    spec = importlib.util.spec_from_file_location("fit_calibration", os.path.join(ROOT, "oracle", "fit_calibration.py"))
    fc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fc)
    recs = fc.features()
    coef, _, _ = fc.validate(recs, n_split=1)
    n50 = S.read_n50(os.path.join(D, "skder_gtdb_results__Concatenated_N50.txt"))
    want = {(a, f): _lines(os.path.join(D, "tc", "skDER_Results_ANI%s_AF%s.txt" % (a, f))) for a in ANI_CUTS for f in AF_CUTS}

    def tv(x):
        return float("%.2f" % float(np.float32(x) * np.float32(100)))

    def identical(ca, cb):
        edges = [(r["a"], r["b"], tv(1.0 - (ca * r["d_cell"] + cb * r["d_span"]) / 100.0), tv(r["af"][0]), tv(r["af"][1])) for r in recs]
        return {k: S.greedy_from_edges(edges, n50, k[0], k[1]) == w for k, w in want.items()}

    shipped = identical(0.53, 0.71)
    assert all(shipped[(a, f)] for a in (99.0, 99.5) for f in AF_CUTS)
    assert sum(shipped.values()) >= 18
    grid = [(round(coef[0] + da, 3), round(coef[1] + db, 3)) for da in (-0.01, -0.005, 0.0, 0.005, 0.01) for db in (-0.01, -0.005, 0.0, 0.005, 0.01)]
    res = {g: identical(*g) for g in grid}
    assert all(r[(99.5, f)] for r in res.values() for f in AF_CUTS)                   # the default cut-off: robust
    holds = [g for g, r in res.items() if all(r[(99.0, f)] for f in AF_CUTS)]
    assert 0 < len(holds) < len(grid)                                                  # knife-edge, on record
    # the reproducing side is the lower-divergence side of the line through the deciding edge
    assert (round(coef[0] - 0.01, 3), round(coef[1] - 0.01, 3)) in holds and (round(coef[0] + 0.01, 3), round(coef[1] + 0.01, 3)) not in holds
