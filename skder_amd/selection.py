"""Representative selection fed IN MEMORY from the engine's edge list.

Host-side counterparts of the reference's down-stream consumers of the skani edge table, with the
reference's exact text conventions so that outputs can be compared file for file:

  genome_information()  ~ src/skDER/skDERsum.cpp:60-165   (connectivity x N50 score, member lists)
  sort_like_coreutils() ~ `sort -k 2 -gr` at src/skDER/skder.py:145-147
  greedy()              ~ src/skDER/skder.py:150-165
  dynamic()             ~ src/skDER/skDERcore.cpp:60-224  (two passes; the CODE's rule, not the README's)
  determine_clusters()  ~ src/skDER/skder.py:168-277      (non-MGE branch)

Edges are (ref, query, ani, af_ref, af_query) with the values ROUNDED TO TWO DECIMALS, i.e. what the
reference would have parsed from skani's text table (both C++ programs `stod` the text).

This module is the READABLE STATEMENT of the rules.  The product path is native: `native_greedy`, `native_dynamic`,
`native_clusters` at the end of the file call skder_amd/csrc/select.cpp through the C ABI on the engine's edge RECORDS (no
per-edge Python, no text round trip); tests/test_selection.py holds the two byte-identical with each other and with the
reference's own binaries."""
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


# ---------------------------------------------------------------------------------------------------------------------------
# native selection (skder_amd/csrc/select.cpp) on engine edge records

def _c_strings(names):
    import ctypes as C
    return (C.c_char_p * max(len(names), 1))(*[n.encode() for n in names])


def _native_args(rows, paths, display):
    import ctypes as C
    import numpy as np
    from .engine import EDGE_DTYPE
    rows = np.ascontiguousarray(rows, dtype=EDGE_DTYPE)
    if display is not None and len(display) != len(paths):
        raise ValueError("display names: one per genome")
    return rows, rows.ctypes.data_as(C.c_void_p), _c_strings(paths), (_c_strings(display) if display is not None else None)


def _enc(x):
    return x.encode() if x else None


def native_greedy(rows, paths: Sequence[str], n50: Sequence[int], min_ani: float, min_af: float, info_txt=None, sorted_txt=None,
                  results_txt=None, display: Sequence[str] = None) -> List[int]:
    """skDERsum + `sort -k 2 -gr` + the greedy loop on edge records; returns the representatives' indices in result-file order"""
    import ctypes as C
    import numpy as np
    from . import _lib
    if len(n50) != len(paths):
        raise ValueError("N50 table: one value per genome")
    rows, prow, cpaths, cdisp = _native_args(rows, paths, display)
    n50a = np.ascontiguousarray(n50, dtype=np.uint64)
    reps = np.zeros(max(len(paths), 1), np.uint32)
    nr = C.c_uint32(0)
    err = C.create_string_buffer(_lib.ERRLEN)
    rc = _lib.lib().skder_amd_select_greedy(prow, len(rows), len(paths), cpaths, n50a.ctypes.data_as(C.POINTER(C.c_uint64)), cdisp,
                                            float(min_ani), float(min_af), _enc(info_txt), _enc(sorted_txt), _enc(results_txt),
                                            reps.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(nr), err, _lib.ERRLEN)
    if rc != 0:
        raise RuntimeError("skder_amd_select_greedy: " + err.value.decode())
    return [int(x) for x in reps[:nr.value]]


def native_dynamic(rows, paths: Sequence[str], n50: Sequence[int], min_ani: float, min_af: float, max_af_diff: float, results_txt=None,
                   display: Sequence[str] = None) -> List[int]:
    """skDERcore on edge records; representatives in N50-file (listing) order"""
    import ctypes as C
    import numpy as np
    from . import _lib
    if len(n50) != len(paths):
        raise ValueError("N50 table: one value per genome")
    rows, prow, cpaths, cdisp = _native_args(rows, paths, display)
    n50a = np.ascontiguousarray(n50, dtype=np.uint64)
    reps = np.zeros(max(len(paths), 1), np.uint32)
    nr = C.c_uint32(0)
    err = C.create_string_buffer(_lib.ERRLEN)
    rc = _lib.lib().skder_amd_select_dynamic(prow, len(rows), len(paths), cpaths, n50a.ctypes.data_as(C.POINTER(C.c_uint64)), cdisp,
                                             float(min_ani), float(min_af), float(max_af_diff), _enc(results_txt),
                                             reps.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(nr), err, _lib.ERRLEN)
    if rc != 0:
        raise RuntimeError("skder_amd_select_dynamic: " + err.value.decode())
    return [int(x) for x in reps[:nr.value]]


def native_clusters(rows, paths: Sequence[str], reps: Sequence[int], af_cutoff: float, ani_cutoff: float, clustering_txt: str,
                    display: Sequence[str] = None) -> None:
    """determineClusters on edge records -> skDER_Clustering.txt"""
    import ctypes as C
    import numpy as np
    from . import _lib
    rows, prow, cpaths, cdisp = _native_args(rows, paths, display)
    ra = np.ascontiguousarray(reps, dtype=np.uint32)
    err = C.create_string_buffer(_lib.ERRLEN)
    rc = _lib.lib().skder_amd_select_clusters(prow, len(rows), len(paths), cpaths, cdisp, ra.ctypes.data_as(C.POINTER(C.c_uint32)), len(ra),
                                              float(af_cutoff), float(ani_cutoff), clustering_txt.encode(), err, _lib.ERRLEN)
    if rc != 0:
        raise RuntimeError("skder_amd_select_clusters: " + err.value.decode())


def rows_from_table(path: str, paths: Sequence[str]):
    """a text table (Skani_Dist_Output.txt of the low_mem_greedy flow, or a golden table) as edge records for the native functions:
    values as fractions whose single-precision percentage prints back to the table's two decimals"""
    import numpy as np
    from .engine import EDGE_DTYPE
    idx = {p: i for i, p in enumerate(paths)}
    ed = edges_from_table(path)
    rows = np.zeros(len(ed), EDGE_DTYPE)
    for k, (r, q, ani, afr, afq) in enumerate(ed):
        rows[k]["ref"] = idx[r]; rows[k]["query"] = idx[q]
        rows[k]["ani"] = np.float32(ani) / np.float32(100); rows[k]["af_ref"] = np.float32(afr) / np.float32(100)
        rows[k]["af_query"] = np.float32(afq) / np.float32(100)
    return rows
