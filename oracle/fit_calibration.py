#!/usr/bin/env python3
"""Fit and VALIDATE the "learned ANI" stand-in (include/skder_amd_spec.h ANI_CAL_CELL / ANI_CAL_SPAN,
ani_oracle.c oracle_model_ani) and the chain padding ANI_PAD against golden table G5.

skani's default output passes its chunk-level k-mer ANI through a gradient-boosted regression whose
model cannot be reconstructed here.  The stand-in is a line through the origin in
two divergences that the engine computes from integer sums of the kept chains:

    d_cell = 100 * (1 - (A / N)^(1/15))     N = all seeds of the cells that hold a kept chain
    d_span = 100 * (1 - (A / S)^(1/15))     S = seeds inside the kept chains' spans
    100 - ANI% = a * d_cell + b * d_span

Two parameters, fitted by least squares to the 561 pairs of G5 (one species, ANI 96.4-100).
`validate()` is the honest part: the fit is repeated on the pairs among a random half of the 34
genomes and scored on the pairs among the OTHER half (no genome in common), 40 splits.  The same
function scores any alternative feature set, which is how the two features were chosen:

    feature set                        in-sample rms / max    held-out rms mean / worst   max mean / worst
    span estimate, 7-knot map (r1)         0.158 / 0.61           0.18 / 0.33                0.63 / 2.37
    cell estimate alone (1 param)          0.168 / 0.59           0.173 / 0.206              0.55 / 0.63
    cell + span (2 params, adopted)        0.142 / 0.43           0.149 / 0.179              0.41 / 0.47
    + log(contigs of chunked genome)*d     0.125 / 0.50           0.132 / 0.169              0.45 / 0.55   (not adopted:
                                           needs a logarithm on the device, larger max error, one species)
    + AF, chains per base, chain length    no held-out gain

Usage:  python oracle/fit_calibration.py        (needs oracle/libani_oracle.so and tests/golden/)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import oracle_py as O  # noqa: E402

G = os.path.join(HERE, "..", "tests", "golden")


def golden_rows():
    rows = {}
    with open(os.path.join(G, "G5_triangle_minaf10_s89.5.tsv")) as f:
        next(f)
        for line in f:
            s = line.rstrip("\n").split("\t")
            rows[(s[0], s[1])] = (float(s[2]), float(s[3]), float(s[4]))
    return rows


def features():
    """one record per G5 pair: names, golden values, the two divergences, span sum, chains, lengths"""
    rows = golden_rows()
    p = O.default_params(learned=0)
    names = sorted({a for a, _ in rows} | {b for _, b in rows})
    gs = {n: O.Genome.load(os.path.join(G, "genomes", n), p) for n in names}
    out = []
    for (a, b), gold in rows.items():
        r = O.pair(gs[a], gs[b], p)
        out.append(dict(a=a, b=b, gold=gold, d_cell=100.0 * (1.0 - r.ani_raw), d_span=100.0 * (1.0 - r.ani_span),
                        span=r.sum_span, chains=r.n_chains, t_ref=gs[a].total_len, t_query=gs[b].total_len,
                        af=(r.af_ref, r.af_query)))
    return out


def validate(recs, cols=("d_cell", "d_span"), n_split=40, seed=3):
    """least squares through the origin on `cols`; returns (coef, in-sample residuals, held-out table)"""
    names = sorted({r["a"] for r in recs} | {r["b"] for r in recs})
    X = np.array([[r[c] for c in cols] for r in recs])
    y = np.array([100.0 - r["gold"][0] for r in recs])
    rng = np.random.default_rng(seed)
    held = []
    for _ in range(n_split):
        perm = rng.permutation(len(names))
        train = {names[i] for i in perm[:len(names) // 2]}
        m_tr = np.array([r["a"] in train and r["b"] in train for r in recs])
        m_te = np.array([r["a"] not in train and r["b"] not in train for r in recs])
        c = np.linalg.lstsq(X[m_tr], y[m_tr], rcond=None)[0]
        res = (X @ c - y)[m_te]
        held.append((np.sqrt((res ** 2).mean()), np.abs(res).max()))
    coef = np.linalg.lstsq(X, y, rcond=None)[0]
    return coef, X @ coef - y, np.array(held)


def fit_pad(recs):
    """chain padding by least squares on the aligned-base counts implied by the uncapped golden AFs"""
    n, gap = [], []
    for r in recs:
        bg = []
        if r["gold"][1] < 99.99:
            bg.append(r["gold"][1] / 100.0 * r["t_ref"])
        if r["gold"][2] < 99.99:
            bg.append(r["gold"][2] / 100.0 * r["t_query"])
        if bg:
            n.append(r["chains"])
            gap.append(np.mean(bg) - r["span"])
    n, gap = np.array(n, float), np.array(gap)
    pad = float((n * gap).sum() / (n * n).sum())
    return pad, gap - pad * n


def main():
    recs = features()
    coef, res, held = validate(recs)
    print("pairs %d   in-sample rms %.3f max %.3f" % (len(recs), np.sqrt((res ** 2).mean()), np.abs(res).max()))
    print("held out (%d splits, disjoint genomes): rms mean %.3f worst %.3f ; max mean %.3f worst %.3f"
          % (len(held), held[:, 0].mean(), held[:, 0].max(), held[:, 1].mean(), held[:, 1].max()))
    print("#define ANI_CAL_CELL %.2f" % coef[0])
    print("#define ANI_CAL_SPAN %.2f" % coef[1])
    for cols in (("d_cell",), ("d_span",)):
        c1, r1, h1 = validate(recs, cols)
        print("  %-8s alone: coef %.3f in-sample rms %.3f held-out rms mean %.3f worst %.3f"
              % (cols[0], c1[0], np.sqrt((r1 ** 2).mean()), h1[:, 0].mean(), h1[:, 0].max()))
    pad, rb = fit_pad(recs)
    print("#define ANI_PAD %d      /* least squares %.1f; aligned-base residual rms %.0f bp */"
          % (round(pad / 10.0) * 10, pad, np.sqrt((rb ** 2).mean())))


if __name__ == "__main__":
    main()
