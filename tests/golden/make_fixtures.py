#!/usr/bin/env python3
"""Regenerate tests/golden/ from the reference checkout (run in the build container only).

The GPU box has no /root/reference, so everything the tests need is committed here as DATA:
  genomes/*.fasta.gz   the 34 gzipped C. granulosum assemblies the reference's own test holds
                       (/root/reference/test_case/skder_gtdb_results/gtdb_ncbi_genomes/); the 7 plain
                       .fna files of test_case/Cutibacterium_granulosum_Genomes_in_GTDB_R214/ are
                       byte-identical (after gunzip) to 7 of these, so they are not stored twice.
  G1..G5 *.tsv         the five golden skani edge tables, directory prefixes of
                       the author's machine stripped from columns 1-2 (basename only).
  downstream/*         golden N50 tables / greedy summaries / result listings / clustering tables,
                       same normalisation.
Nothing here is reference source code; only inputs and expected outputs.
"""
import gzip
import hashlib
import os
import re
import shutil
import sys

REF = "/root/reference/test_case"
HERE = os.path.dirname(os.path.abspath(__file__))


def strip_paths(line: str) -> str:
    # any absolute path token -> its basename
    return re.sub(r"/Users/[^\t\n;,]*/", "", line)


def norm_file(src: str, dst: str) -> None:
    with open(src) as f, open(dst, "w") as o:
        for line in f:
            o.write(strip_paths(line))


def main() -> None:
    if not os.path.isdir(REF):
        sys.exit("reference checkout not present; fixtures are already committed")
    gdir = os.path.join(HERE, "genomes")
    os.makedirs(gdir, exist_ok=True)
    src_g = os.path.join(REF, "skder_gtdb_results", "gtdb_ncbi_genomes")
    for fn in sorted(os.listdir(src_g)):
        shutil.copyfile(os.path.join(src_g, fn), os.path.join(gdir, fn))

    # map of the 7 plain-FASTA basenames (G1/G2/G3/G4 tables) -> the gz fixture with the same bytes
    plain_dir = os.path.join(REF, "Cutibacterium_granulosum_Genomes_in_GTDB_R214")
    gz_md5 = {}
    for fn in sorted(os.listdir(gdir)):
        with gzip.open(os.path.join(gdir, fn), "rb") as f:
            gz_md5[hashlib.md5(f.read()).hexdigest()] = fn
    with open(os.path.join(HERE, "plain_to_gz.tsv"), "w") as o:
        for fn in sorted(os.listdir(plain_dir)):
            with open(os.path.join(plain_dir, fn), "rb") as f:
                o.write("%s\t%s\n" % (fn, gz_md5[hashlib.md5(f.read()).hexdigest()]))

    tables = {
        "G1_triangle_minaf50_s89.tsv": "skder_results/Skani_Triangle_Edge_Output.txt",
        "G2_triangle_old_minaf90.tsv": "expected_skder_results/Skani_Triangle_Edge_Output.txt",
        "G3_triangle_old.tsv": "expected_cidder_results/Skani_Triangle_Edge_Output.txt",
        "G4_dist.tsv": "cidder_results/skani_for_Clustering_Workspace/Skani_Dist_Output.txt",
        "G4_dist_reps.txt": "cidder_results/skani_for_Clustering_Workspace/Reps_Listing.txt",
        "G4_dist_nonreps.txt": "cidder_results/skani_for_Clustering_Workspace/NonReps_Listing.txt",
        "G5_triangle_minaf10_s89.5.tsv": "skder_gtdb_results/Skani_Triangle_Edge_Output.txt",
    }
    for dst, src in tables.items():
        norm_file(os.path.join(REF, src), os.path.join(HERE, dst))

    ddir = os.path.join(HERE, "downstream")
    os.makedirs(ddir, exist_ok=True)
    for run in ("skder_results", "skder_gtdb_results"):
        for fn in ("All_Genomes_Listing.txt", "Concatenated_N50.txt",
                   "Genome_Information_for_Greedy_Clustering.txt",
                   "Genome_Information_for_Greedy_Clustering.sorted.txt",
                   "skDER_Results.txt", "skDER_Clustering.txt"):
            p = os.path.join(REF, run, fn)
            if os.path.isfile(p):
                norm_file(p, os.path.join(ddir, run + "__" + fn))
    tc = os.path.join(REF, "skder_gtdb_results", "skDER_Result")
    os.makedirs(os.path.join(ddir, "tc"), exist_ok=True)
    for fn in sorted(os.listdir(tc)):
        norm_file(os.path.join(tc, fn), os.path.join(ddir, "tc", fn))


if __name__ == "__main__":
    main()
