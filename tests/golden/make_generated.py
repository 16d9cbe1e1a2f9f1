#!/usr/bin/env python3
"""Goldens the reference's test run does NOT hold, generated here FROM the reference itself (build container only):

  * dynamic mode: stdout of the reference's skDERcore, compiled by oracle/build_ref.sh into oracle/_ref/ from
    /root/reference/src/skDER/skDERcore.cpp, on golden tables G1 and G5 for a grid of (ANI, AF, max AF difference);
  * secondary clustering: /root/reference/src/skDER/skder.py determineClusters (imported, PYTHONDONTWRITEBYTECODE=1, stub
    modules for the missing Bio / aiofile packages) on the greedy and the dynamic listings of G1 and G5.

Only inputs (the committed, path-normalised golden tables) and outputs are stored: tests/golden/downstream/generated/.
Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_generated.py
"""
import os
import subprocess
import sys
import tempfile
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_SRC = "/root/reference/src"
CORE = os.path.join(ROOT, "oracle", "_ref", "skDERcore")
SUM = os.path.join(ROOT, "oracle", "_ref", "skDERsum")
OUT = os.path.join(HERE, "downstream", "generated")

TABLES = {"G1": ("G1_triangle_minaf50_s89.tsv", "skder_results__Concatenated_N50.txt"),
          "G5": ("G5_triangle_minaf10_s89.5.tsv", "skder_gtdb_results__Concatenated_N50.txt")}
DYNAMIC_GRID = [(99.5, 50.0, 10.0), (99.0, 50.0, 10.0), (99.0, 90.0, 10.0), (98.0, 50.0, 10.0), (97.0, 25.0, 5.0), (99.0, 50.0, 0.0)]
CLUSTER_GRID = [(99.5, 50.0), (99.0, 50.0), (98.0, 25.0)]


def import_reference():
    if not os.environ.get("PYTHONDONTWRITEBYTECODE"):
        sys.exit("run with PYTHONDONTWRITEBYTECODE=1 (the reference checkout must stay untouched)")
    for name in ("Bio", "Bio.SeqIO", "aiofile"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["Bio"].SeqIO = sys.modules["Bio.SeqIO"]
    sys.path.insert(0, REF_SRC)
    from skDER import skder as ref_skder        # noqa: E402
    return ref_skder


def main():
    if not (os.path.isfile(CORE) and os.path.isdir(REF_SRC)):
        sys.exit("needs /root/reference and oracle/_ref (bash oracle/build_ref.sh): the generated fixtures are already committed")
    ref = import_reference()
    os.makedirs(OUT, exist_ok=True)
    for tag, (table, n50) in TABLES.items():
        tp, nf = os.path.join(HERE, table), os.path.join(HERE, "downstream", n50)
        listings = {}
        for ani, af, maxd in DYNAMIC_GRID:
            txt = subprocess.run([CORE, tp, nf, str(ani), str(af), str(maxd)], capture_output=True, text=True, check=True).stdout
            name = "dynamic__%s__ANI%s_AF%s_D%s.txt" % (tag, ani, af, maxd)
            open(os.path.join(OUT, name), "w").write(txt)
            listings[("dynamic", ani, af, maxd)] = txt
        with tempfile.TemporaryDirectory() as td:
            for ani, af in CLUSTER_GRID:
                # greedy listing by the reference's own chain: skDERsum | sort -k 2 -gr | greedy loop (skder.py:136-165)
                info = os.path.join(td, "info.txt")
                with open(info, "w") as o:
                    subprocess.run([SUM, tp, nf, str(ani), str(af)], stdout=o, check=True)
                srt = os.path.join(td, "info.sorted.txt")
                with open(srt, "w") as o:
                    subprocess.run(["sort", "-k", "2", "-gr", info], stdout=o, check=True, env=dict(os.environ, LC_ALL="C"))
                reps, seen = [], set()
                for line in open(srt):
                    c = line.rstrip("\n").split("\t")
                    if c[0] in seen:
                        continue
                    reps.append(c[0])
                    seen.add(c[0])
                    if len(c) > 2 and c[2].strip():
                        seen.update(x.strip() for x in c[2].split("; "))
                for mode, rlist in (("greedy", reps), ("dynamic", listings[("dynamic", ani, af, 10.0)].split()) if (("dynamic", ani, af, 10.0) in listings) else ("greedy", reps)):
                    rf = os.path.join(td, "reps.txt")
                    open(rf, "w").write("".join(r + "\n" for r in rlist))
                    cf = os.path.join(OUT, "clusters__%s__%s__ANI%s_AF%s.txt" % (tag, mode, ani, af))
                    ref.determineClusters(rf, tp, None, None, cf, af, ani)
                    open(os.path.join(OUT, "reps__%s__%s__ANI%s_AF%s.txt" % (tag, mode, ani, af)), "w").write("".join(r + "\n" for r in rlist))
    print("wrote", len(os.listdir(OUT)), "files to", OUT)


if __name__ == "__main__":
    main()
