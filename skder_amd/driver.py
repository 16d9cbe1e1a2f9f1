"""End-to-end dereplication flow of bin/skder on top of the MI355X engine:
listing -> N50 table -> edge table (GPU) -> representative selection -> result files.

Only the flow around the hot path is reproduced (no downloads, MGE filtering, plots); file names and
formats follow the reference so that outputs can be diffed against its result directories.

    python -m skder_amd.driver -g GENOME_DIR_OR_FILES... -o OUT/ [-d greedy|dynamic|low_mem_greedy]
                               [-i 99.5] [-f 50.0] [-a 10.0] [-n] [-l] [-p "-s 89.5"] [--store FILE] [--devices 0,1,..]

The FASTA files are read ONCE: the pass that uploads them also yields Concatenated_N50.txt
(SURVEY.md 8f-2), the edge rows reach the selection step in memory (8f-1; the text table is still
written because it is one of skDER's outputs) and, with --store, the sketches are kept on disk so a
later run over the same listing skips ingest (8f-4).
"""
import argparse
import json
import os
import shutil
import sys
from collections import OrderedDict

from . import selection
from .skder import Database, lowMemGreedyDerep, parse_skani_params, runSkaniDist

ACCEPTED_SUFFICES = ("fasta", "fas", "fna", "fa")      # util.py:21


def list_genomes(inputs):
    out = []
    for x in inputs:
        if os.path.isdir(x):
            for fn in os.listdir(x):             # os.listdir order, as util.processInputGenomes
                base = fn[:-3] if fn.endswith(".gz") else fn
                if base.split(".")[-1] in ACCEPTED_SUFFICES:
                    out.append(os.path.abspath(os.path.join(x, fn)))
        else:
            out.append(os.path.abspath(x))
    return out


def _file_stamps(genomes):
    """[path, size, mtime in ns] of every FASTA file: what a sketch store was built from"""
    out = []
    for g in genomes:
        try:
            st = os.stat(g)
            out.append([g, st.st_size, st.st_mtime_ns])
        except OSError:
            out.append([g, None, None])          # gone: nothing to compare with
    return out


def _store_is_fresh(stamps_then, stamps_now):
    """same paths in the same order; every file that still exists has the size and modification time it had when the store
    was written (a file that is gone cannot contradict the store: skani's sketch directory does not need the FASTA files
    afterwards either)"""
    if len(stamps_then) != len(stamps_now):
        return False
    for then, now in zip(stamps_then, stamps_now):
        if then[0] != now[0]:
            return False
        if now[1] is not None and (then[1], then[2]) != (now[1], now[2]):
            return False
    return True


def open_database(listing, genomes, n50_file, store=None, devices=None):
    """the resident sketch database of the listing + its N50 table; reuses a sketch store when it
    describes exactly these files: same paths in the same order AND unchanged size and modification
    time of every file (a store does not notice a FASTA file replaced in place by itself)"""
    stamps_file = store + ".files.json" if store else None
    if store and os.path.isfile(store) and os.path.isfile(stamps_file):
        try:
            fresh = _store_is_fresh(json.load(open(stamps_file)), _file_stamps(genomes))
        except (OSError, ValueError):
            fresh = False
        if fresh:
            db = Database.load(store)
            if db.paths == list(genomes):
                with open(n50_file, "w") as f:
                    f.write("".join("%s\t%d\n" % kv for kv in zip(db.paths, db.n50)))
                return db
            db.close()
        else:
            sys.stderr.write("sketch store %s is stale (files changed since it was written): re-reading the FASTA files\n" % store)
    stamps = _file_stamps(genomes) if store else None
    db = Database.from_listing(listing, n50_file, devices=devices)
    if store:
        db.save(store)
        with open(stamps_file, "w") as f:
            json.dump(stamps, f)
    return db


def run(genomes, outdir, mode="greedy", ani=99.5, af=50.0, max_af_dist=10.0, clusters=False, params="-s X", store=None,
        symlink=False, devices=None, name_map=None):
    outdir = os.path.abspath(outdir) + "/"
    os.makedirs(outdir, exist_ok=True)
    if params == "-s X":                          # bin/skder:199-201
        params = "-s %s" % max(ani - 10.0, 0.0)
    if clusters and mode == "dynamic":            # bin/skder:216-219
        af_tri = max(af - 20.0, 0.0)
    else:
        af_tri = af
    listing = outdir + "All_Genomes_Listing.txt"
    with open(listing, "w") as f:
        f.write("".join(g + "\n" for g in genomes))
    n50_file = outdir + "Concatenated_N50.txt"
    db = open_database(listing, genomes, n50_file, store, devices)
    try:
        return _run(db, genomes, outdir, mode, ani, af, af_tri, max_af_dist, clusters, params, listing, n50_file, symlink, name_map)
    finally:
        db.close()


PRESELECTED_ANI_CUTOFFS = [90.0, 95.0, 97.0, 98.0, 99.0, 99.5]         # bin/skder:54-55
PRESELECTED_AF_CUTOFFS = [10.0, 25.0, 50.0, 75.0, 90.0]


def run_test_cutoffs(genomes, outdir, mode="greedy", max_af_dist=10.0, params="-s X", ani=99.5, store=None, devices=None, name_map=None):
    """bin/skder -tc (bin/skder:331-401): ONE edge table at --min-af 10 (0 for the dynamic mode: skder.py:19-25), then the selection at every
    pre-selected (ANI, AF) cut-off pair: skDER_Result/skDER_Results_ANI<a>_AF<f>.txt and the counts the reference prints and plots
    (Parameter_Impacts_Overview.tsv here; the PDF heat map is the control plane's).  The table is computed once and stays in memory;
    low_mem_greedy runs its searches per cut-off pair on the resident database, as the reference does on its sketch directory.
    Returns {(ani, af): number of representatives}."""
    outdir = os.path.abspath(outdir) + "/"
    os.makedirs(outdir + "skDER_Result/", exist_ok=True)
    if params == "-s X":
        params = "-s %s" % max(ani - 10.0, 0.0)
    listing = outdir + "All_Genomes_Listing.txt"
    with open(listing, "w") as f:
        f.write("".join(g + "\n" for g in genomes))
    n50_file = outdir + "Concatenated_N50.txt"
    db = open_database(listing, genomes, n50_file, store, devices)
    counts = OrderedDict()
    try:
        shown = [name_map[p] for p in db.paths] if name_map else None
        rows = None
        if mode in ("greedy", "dynamic"):
            min_af = 10.0 if mode == "greedy" else 0.0
            rows = db.triangle(min_af, parse_skani_params(params), out_tsv=outdir + "Skani_Triangle_Edge_Output.txt")
        elif mode != "low_mem_greedy":
            raise ValueError("unknown dereplication mode " + mode)
        n50v = list(db.n50)
        for a in PRESELECTED_ANI_CUTOFFS:
            for f_ in PRESELECTED_AF_CUTOFFS:
                res = outdir + "skDER_Result/skDER_Results_ANI%s_AF%s.txt" % (a, f_)
                if mode == "greedy":
                    n = len(selection.native_greedy(rows, db.paths, n50v, a, f_, outdir + "Genome_Information_for_Greedy_Clustering.txt",
                                                    outdir + "Genome_Information_for_Greedy_Clustering.sorted.txt", res, display=shown))
                elif mode == "dynamic":
                    n = len(selection.native_dynamic(rows, db.paths, n50v, a, f_, max_af_dist, res, display=shown))
                else:
                    ws = outdir + "skDER_iterative_greedy_workspace/"
                    os.makedirs(ws, exist_ok=True)
                    lowMemGreedyDerep(listing, ws, n50_file, res, outdir, a, f_, None, mge_proc_to_unproc_mapping=name_map, database=db)
                    n = sum(1 for _ in open(res))
                counts[(a, f_)] = n
        with open(outdir + "Parameter_Impacts_Overview.tsv", "w") as f:
            f.write("ANI/AF\t" + "\t".join(str(x) for x in PRESELECTED_AF_CUTOFFS) + "\n")
            for a in PRESELECTED_ANI_CUTOFFS:
                f.write(str(a) + "\t" + "\t".join(str(counts[(a, x)]) for x in PRESELECTED_AF_CUTOFFS) + "\n")
    finally:
        db.close()
    return counts


def _run(db, genomes, outdir, mode, ani, af, af_tri, max_af_dist, clusters, params, listing, n50_file, symlink=False, name_map=None):
    n50 = OrderedDict(zip(db.paths, db.n50))
    # name_map: the reference's mge_proc_to_unproc_mapping (skder.py:76-92, 127-129, 160-163, 236-253) -- the genomes were
    # pre-processed (MGE regions cut out: out of scope here) and the result files name the originals
    shown = [name_map[p] for p in db.paths] if name_map else None
    result_file = outdir + "skDER_Results.txt"
    edge_file = outdir + "Skani_Triangle_Edge_Output.txt"
    if mode == "low_mem_greedy":
        ws = outdir + "skDER_iterative_greedy_workspace/"      # bin/skder:438
        os.makedirs(ws, exist_ok=True)
        lowMemGreedyDerep(listing, ws, n50_file, result_file, outdir, ani, af, None, database=db)
        if clusters:
            cdir = outdir + "skani_dist_Workspace/"              # bin/skder:466
            os.makedirs(cdir, exist_ok=True)
            edge_file = outdir + "Skani_Dist_Output.txt"
            runSkaniDist(cdir, result_file, listing, edge_file, params, af, mode, False, None)
        reps = [l.strip() for l in open(result_file)]
        if shown is not None:                      # skder.py:130-132: the names written are the mapped ones; the flow goes on with the listing's
            with open(result_file, "w") as f:
                f.write("".join(name_map[r] + "\n" for r in reps))
    else:
        # the edge rows reach the selection in memory; the selection itself is native (skder_amd/csrc/select.cpp): no per-edge Python
        rows = db.triangle(af_tri, parse_skani_params(params), out_tsv=edge_file)
        n50v = [n50[p] for p in db.paths]
        if mode == "greedy":
            rep_idx = selection.native_greedy(rows, db.paths, n50v, ani, af, outdir + "Genome_Information_for_Greedy_Clustering.txt",
                                              outdir + "Genome_Information_for_Greedy_Clustering.sorted.txt", result_file, display=shown)
        elif mode == "dynamic":
            rep_idx = selection.native_dynamic(rows, db.paths, n50v, ani, af, max_af_dist, result_file, display=shown)
        else:
            raise ValueError("unknown dereplication mode " + mode)
        reps = [db.paths[i] for i in rep_idx]
    if clusters:
        if mode == "low_mem_greedy":
            rows = selection.rows_from_table(edge_file, db.paths)
            idx = {p: i for i, p in enumerate(db.paths)}
            rep_idx = [idx[r] for r in reps]
        selection.native_clusters(rows, db.paths, rep_idx, af, ani, outdir + "skDER_Clustering.txt", display=shown)
    rep_dir = outdir + "Dereplicated_Representative_Genomes/"
    os.makedirs(rep_dir, exist_ok=True)
    for r in reps:                                  # util.py:411-427: copies unless -l / --symlink was given
        r = name_map[r] if name_map else r          # (the originals, as the result file names them)
        dst = rep_dir + os.path.basename(r)
        if os.path.lexists(dst):
            continue
        try:
            if symlink:
                os.symlink(r, dst)
            else:
                shutil.copy2(r, rep_dir)
        except OSError:                               # util.py:425-427: a warning, not an error
            sys.stderr.write('Warning: issues copying over representative genome %s to final dereplicated sub-directory.\n' % r)
    with open(outdir + "COMPLETED.txt", "w") as f:
        f.write("skDER completed successfully!\n")
    return reps


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawTextHelpFormatter)
    ap.add_argument("-g", "--genomes", nargs="+", required=True)
    ap.add_argument("-o", "--output-directory", required=True)
    ap.add_argument("-d", "--dereplication-mode", default="greedy")
    ap.add_argument("-i", "--percent-identity-cutoff", type=float, default=99.5)
    ap.add_argument("-f", "--aligned-fraction-cutoff", type=float, default=50.0)
    ap.add_argument("-a", "--max-af-distance-cutoff", type=float, default=10.0)
    ap.add_argument("-p", "--skani-triangle-parameters", default="-s X")
    ap.add_argument("-n", "--determine-clusters", action="store_true")
    ap.add_argument("-l", "--symlink", action="store_true", help="symlink the representatives instead of copying them (bin/skder:106)")
    ap.add_argument("--name-map", default=None, help="TSV `listed path<TAB>name to report`: the reference's mge_proc_to_unproc_mapping "
                                                     "(genomes pre-processed elsewhere; result files and the representatives' directory use the originals)")
    ap.add_argument("--store", default=None, help="sketch store file: loaded if present and still describing these files, written otherwise")
    ap.add_argument("--devices", default=None, help="comma-separated GPU indices (default: $SKDER_AMD_DEVICE or 0): with several, the genomes "
                                                    "are sketched in shares, the sketches exchanged between the GPUs and the pair matrix dealt out by rows")
    ap.add_argument("-tc", "--test-cutoffs", action="store_true", help="one edge table, the selection at every pre-selected cut-off pair (bin/skder:99)")
    ap.add_argument("--ani", choices=("model", "raw"), default="model",
                    help="which ANI the tables carry and the selection reads: `model` = after the learned-ANI stand-in (what skani prints by default), "
                         "`raw` = the chunk-level k-mer estimate, skani's --no-learned-ani (for simulated genomes with independent substitutions, "
                         "where the stand-in reads 1.24 x the true divergence: DESIGN.md section 2)")
    a = ap.parse_args(argv)
    from . import _lib
    if _lib.lib().skder_amd_set_ani_output(1 if a.ani == "raw" else 0) < 0:
        raise RuntimeError("skder_amd_set_ani_output failed")
    name_map = None
    if a.name_map:
        with open(a.name_map) as f:
            name_map = dict(l.rstrip("\n").split("\t")[:2] for l in f if l.strip())
    if a.test_cutoffs:
        counts = run_test_cutoffs(list_genomes(a.genomes), a.output_directory, a.dereplication_mode, a.max_af_distance_cutoff,
                                  a.skani_triangle_parameters, a.percent_identity_cutoff, a.store,
                                  [int(x) for x in a.devices.split(",")] if a.devices else None, name_map)
        print("Number of representative genomes selected:")
        print(open(os.path.join(a.output_directory, "Parameter_Impacts_Overview.tsv")).read())
        return counts
    reps = run(list_genomes(a.genomes), a.output_directory, a.dereplication_mode, a.percent_identity_cutoff,
               a.aligned_fraction_cutoff, a.max_af_distance_cutoff, a.determine_clusters, a.skani_triangle_parameters, a.store,
               a.symlink, [int(x) for x in a.devices.split(",")] if a.devices else None, name_map)
    print("%d representative genomes -> %s" % (len(reps), os.path.join(a.output_directory, "skDER_Results.txt")))


if __name__ == "__main__":
    main()
