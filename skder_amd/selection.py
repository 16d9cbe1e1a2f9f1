"""Representative selection fed IN MEMORY from the engine's edge list.

Host-side counterparts of the reference's down-stream consumers of the skani edge table, with the
reference's exact text conventions so that outputs can be compared file for file:

  genome_information()  ~ src/skDER/skDERsum.cpp:60-165   (connectivity x N50 score, member lists)
  sort_like_coreutils() ~ `sort -k 2 -gr` at src/skDER/skder.py:145-147
  greedy()              ~ src/skDER/skder.py:150-165
  dynamic()             ~ src/skDER/skDERcore.cpp:60-224  (two passes; the CODE's rule, not the README's)
  determine_clusters()  ~ src/skDER/skder.py:168-277      (non-MGE branch)

Edges are (ref, query, ani, af_ref, af_query) with the values ROUNDED TO TWO DECIMALS, i.e. what the
reference would have parsed from skani's text table (both C++ programs `stod` the text)."""
from collections import OrderedDict
from typing import Dict, Iterable, List, Sequence, Tuple

Edge = Tuple[str, str, float, float, float]


def edges_from_table(path: str) -> List[Edge]:
    out = []
    with open(path) as f:
        next(f)
        for line in f:
            if not line.strip():
                continue
            s = line.rstrip("\n").split("\t")
            out.append((s[0], s[1], float(s[2]), float(s[3]), float(s[4])))
    return out


def edges_from_engine(edges, paths: Sequence[str]) -> List[Edge]:
    """engine edge records (skder_amd.engine.EDGE_DTYPE) -> text-precision edges"""
    import numpy as np
    out = []
    for e in edges:
        ani, afr, afq = np.float32(e["ani"]), np.float32(e["af_ref"]), np.float32(e["af_query"])
        out.append((paths[int(e["ref"])], paths[int(e["query"])], float("%.2f" % (ani * np.float32(100))),
                    float("%.2f" % (afr * np.float32(100))), float("%.2f" % (afq * np.float32(100)))))
    return out


def _fmt_score(x: float) -> str:
    # C++ `ostream << double` with default precision: %g with 6 significant digits
    return "%g" % x


def genome_information(edges: Iterable[Edge], n50: "OrderedDict[str, int]", min_ani: float, min_af: float) -> List[str]:
    """lines of Genome_Information_for_Greedy_Clustering.txt (skDERsum.cpp:90-165)"""
    conn: Dict[str, int] = {}
    members: Dict[str, List[str]] = {}
    for q, s, ani, q_af, s_af in edges:      # skDERsum names column 1 "query" and column 2 "subject"
        if ani >= min_ani and (q_af >= min_af or s_af >= min_af):
            if s_af >= min_af:
                conn[q] = conn.get(q, 0) + 1
                members.setdefault(q, []).append(s)
            if q_af >= min_af:
                conn[s] = conn.get(s, 0) + 1
                members.setdefault(s, []).append(q)
    lines = []
    for sample, n in n50.items():
        if sample in conn:
            lines.append(sample + "\t" + _fmt_score(float(n) * float(conn[sample])) + "\t" + "; ".join(members[sample]))
        else:
            lines.append(sample + "\t0.0\t")
    return lines


def sort_like_coreutils(lines: List[str]) -> List[str]:
    """`sort -k 2 -gr` in the C locale: key = field 2 to end of line compared as a general number,
    descending; ties by the whole line, bytes, also descending (the -r applies to the last resort)."""
    def key(line: str):
        rest = line.split("\t", 1)[1] if "\t" in line else ""
        tok = rest.lstrip().split()[0] if rest.strip() else ""
        try:
            v = float(tok)
        except ValueError:
            v = float("-inf")
        return (v, line.encode())
    return sorted(lines, key=key, reverse=True)


def greedy(sorted_lines: List[str]) -> List[str]:
    """skder.py:150-165"""
    reps, accounted = [], set()
    for line in sorted_lines:
        ls = line.strip("\n").split("\t")
        if ls[0] in accounted:
            continue
        for g in ls[2].split("; "):
            accounted.add(g)
        reps.append(ls[0])
    return reps


def greedy_from_edges(edges, n50, min_ani, min_af) -> List[str]:
    return greedy(sort_like_coreutils(genome_information(edges, n50, min_ani, min_af)))


def dynamic(edges: Iterable[Edge], n50: "OrderedDict[str, int]", min_ani: float, min_af: float, max_af_diff: float) -> List[str]:
    """skDERcore.cpp: connectivity pass (:95-98), then per edge: if af_query - af_subject <= max_af_diff
    the genome with the LARGER AF is redundant (ties: subject), else the lower N50*connectivity score is
    (ties: subject) (:169-186); survivors in N50-file order (:200-216)."""
    edges = list(edges)
    conn: Dict[str, int] = {}
    for q, s, ani, af_q, af_s in edges:
        if ani >= min_ani and (af_q >= min_af or af_s >= min_af):
            conn[q] = conn.get(q, 0) + 1
            conn[s] = conn.get(s, 0) + 1
    redundant = set()
    for q, s, ani, af_q, af_s in edges:
        if ani >= min_ani and (af_q >= min_af or af_s >= min_af):
            if af_q - af_s <= max_af_diff:
                redundant.add(q if af_q > af_s else s)
            else:
                qs = float(n50.get(q, 0)) * float(conn.get(q, 0))
                ss = float(n50.get(s, 0)) * float(conn.get(s, 0))
                redundant.add(s if qs >= ss else q)
    return [g for g in n50 if g not in redundant]


def determine_clusters(reps: Sequence[str], edges: Iterable[Edge], af_cutoff: float, ani_cutoff: float) -> List[str]:
    """lines of skDER_Clustering.txt (skder.py:168-277, branch without MGE mapping)"""
    out = ["genome\tnearest_representative_genome\taverage_nucleotide_identity\talignment_fraction\tmatch_category"]
    rep_set = set(reps)
    for r in reps:
        out.append(r + "\t" + r + "\t100.0\t100.0\trepresentative_to_self")
    strict: "OrderedDict[str, list]" = OrderedDict()
    loose: "OrderedDict[str, list]" = OrderedDict()

    def upd(table, g, other, ani, af):
        cur = table.setdefault(g, [["NA"], 0.0, 0.0])      # python sets of one element print alike; keep insertion order
        if ani > cur[1]:
            table[g] = [[other], ani, af]
        elif ani == cur[1]:
            if af > cur[2]:
                table[g] = [[other], ani, af]
            elif af == cur[2]:
                if other not in cur[0]:
                    cur[0].append(other)

    for ref, que, ani, raf, qaf in edges:
        if que in rep_set and ref not in rep_set:
            upd(strict if raf >= af_cutoff else loose, ref, que, ani, raf)
            # the reference touches the default-dict of the other table only on assignment, so no entry there
        if ref in rep_set and que not in rep_set:
            upd(strict if qaf >= af_cutoff else loose, que, ref, ani, qaf)
    found = set()
    for g, (names, ani, af) in strict.items():
        found.add(g)
        cat = "within_cutoffs_requested" if ani >= ani_cutoff else "outside_cutoffs_requested"
        out.append("\t".join([g, ", ".join(names), str(ani), str(af), cat]))
    for g, (names, ani, af) in loose.items():
        if g in found:
            continue
        out.append("\t".join([g, ", ".join(names), str(ani), str(af), "outside_cutoffs_requested"]))
    return out


def read_n50(path: str) -> "OrderedDict[str, int]":
    d: "OrderedDict[str, int]" = OrderedDict()
    with open(path) as f:
        for line in f:
            line = line.strip()
            if line:
                g, n = line.split("\t")
                d[g] = int(float(n))
    return d
