#!/usr/bin/env python3
"""Fit the "learned ANI" stand-in (ani_oracle.c calibrate_ani / include/skder_amd_spec.h ANI_CAL_*).

skani's default output passes its chained k-mer ANI through a gradient-boosted regression whose
model file cannot be reconstructed here.  The stand-in is a piecewise-linear
map  d_out = f(d_raw),  d = 100 - ANI%,  with fixed knots; the knot values are least-squares
fitted to golden table G5 (561 pairs, one species, ANI 96.4-100) and printed as a C initialiser.
Outside the fitted range the map continues with slope 1 (a pure shift): unpinned.

Usage:  python oracle/fit_calibration.py        (needs oracle/libani_oracle.so and tests/golden/)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import oracle_py as O  # noqa: E402

G = os.path.join(HERE, "..", "tests", "golden")
KNOTS = [0.0, 0.1, 0.5, 1.0, 1.5, 2.0, 2.5]


def main():
    rows = {}
    with open(os.path.join(G, "G5_triangle_minaf10_s89.5.tsv")) as f:
        next(f)
        for line in f:
            s = line.rstrip("\n").split("\t")
            rows[(s[0], s[1])] = float(s[2])
    p = O.default_params(learned=0)
    names = sorted({a for a, _ in rows} | {b for _, b in rows})
    gs = {n: O.Genome.load(os.path.join(G, "genomes", n), p) for n in names}
    x, y = [], []
    for (a, b), ani in rows.items():
        r = O.pair(gs[a], gs[b], p)
        x.append(100.0 * (1.0 - r.ani_raw))
        y.append(100.0 - ani)
    x, y = np.array(x), np.array(y)
    k = np.array(KNOTS)
    # hat basis; value at knot 0 fixed to 0; beyond the last knot: y_last + (d - k_last)
    A = np.zeros((len(x), len(k) - 1))
    rhs = y.copy()
    for n, d in enumerate(x):
        if d >= k[-1]:
            A[n, -1] = 1.0
            rhs[n] -= d - k[-1]
            continue
        i = np.searchsorted(k, d, side="right") - 1
        t = (d - k[i]) / (k[i + 1] - k[i])
        if i >= 1:
            A[n, i - 1] += 1 - t
        A[n, i] += t
    coef = np.linalg.lstsq(A, rhs, rcond=None)[0]
    vals = np.concatenate([[0.0], coef])
    pred = np.interp(np.minimum(x, k[-1]), k, vals) + np.maximum(x - k[-1], 0)
    res = pred - y
    print("pairs %d  rms %.4f  max %.4f" % (len(x), np.sqrt((res ** 2).mean()), np.abs(res).max()))
    print("#define ANI_CAL_N %d" % len(k))
    print("#define ANI_CAL_X {" + ", ".join("%.2f" % v for v in k) + "}")
    print("#define ANI_CAL_Y {" + ", ".join("%.4f" % v for v in vals) + "}")


if __name__ == "__main__":
    main()
