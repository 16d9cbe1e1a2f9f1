"""Probe for the REAL reference engine: the `skani` binary that raufs/skDER shells out to
(/root/reference/src/skDER/skder.py:16-26, 58-61, 103, 119).  TEST / MEASUREMENT INFRASTRUCTURE ONLY,
like everything under oracle/: imported by tests/ and by bench.py's parity and cpu_baseline legs, never by skder_amd.

skani is a third-party Rust program, version unpinned by the reference (skDER_env.yml:12) and absent
from the build container and from the GPU boxes this repository has run on so far.  The day a box has
it on PATH, bench.py takes the only true numbers there are without a code change:

  * parity      `skani triangle` on the same listing as the drop-in, cell by cell (max |dANI|, |dAF|);
  * cpu_baseline `skani triangle -t $(nproc)` wall-clocked on the bench's bounded sample,
                 reported with kind "reference" and `skani -V`.

When it is absent every function returns None and the callers say "skani unavailable" (SURVEY.md 8d)."""
import os
import shutil
import subprocess
import time


def _candidates():
    """where a conda / mamba install of skDER's environment (skDER_env.yml:12) would put the binary, beyond PATH"""
    import glob
    home = os.path.expanduser("~")
    pre = [os.environ.get("CONDA_PREFIX"), os.environ.get("MAMBA_ROOT_PREFIX")]
    dirs = [os.path.join(p, "bin") for p in pre if p]
    for root in (os.path.join(home, "miniconda3"), os.path.join(home, "miniforge3"), os.path.join(home, "mambaforge"), os.path.join(home, "anaconda3"),
                 os.path.join(home, "micromamba"), "/opt/conda", "/opt/miniconda3", "/opt/miniforge3", "/usr/local/conda"):
        dirs.append(os.path.join(root, "bin"))
        dirs += sorted(glob.glob(os.path.join(root, "envs", "*", "bin")))
    dirs += [os.path.join(home, ".cargo", "bin"), "/usr/local/bin"]
    return dirs


def find():
    """absolute path of the skani executable, or None: PATH first, then the usual conda / cargo locations
    (SKANI_REF_NO_SEARCH=1 restricts the probe to PATH: the tests' "absent" state)"""
    exe = shutil.which("skani")
    if exe or os.environ.get("SKANI_REF_NO_SEARCH") == "1":
        return exe
    for d in _candidates():
        q = os.path.join(d, "skani")
        if os.path.isfile(q) and os.access(q, os.X_OK):
            return q
    return None


def version(exe=None):
    exe = exe or find()
    if not exe:
        return None
    try:
        r = subprocess.run([exe, "-V"], capture_output=True, text=True, timeout=60)
    except (OSError, subprocess.SubprocessError):
        return None
    return (r.stdout.strip() or r.stderr.strip()) or None


def read_table(path):
    """skani's 7-column table -> {frozenset(basenames): (ani, af_ref, af_query, basename of Ref_file)}"""
    rows = {}
    with open(path) as f:
        next(f, None)
        for line in f:
            c = line.rstrip("\n").split("\t")
            if len(c) < 5:
                continue
            a, b = os.path.basename(c[0]), os.path.basename(c[1])
            rows[frozenset((a, b))] = (float(c[2]), float(c[3]), float(c[4]), a)
    return rows


def triangle(listing, out_tsv, min_af, screen, threads):
    """the command line of skder.py:16-18: `skani triangle -l L --min-af A -E -s S -t T -o OUT`.
    Returns {"seconds", "version", "command", "rows"} or None when skani is not installed; raises if it is
    installed and fails (as util.runCmd does when the output file is missing, util.py:636-652)."""
    exe = find()
    if not exe:
        return None
    cmd = [exe, "triangle", "-l", listing, "--min-af", str(min_af), "-E", "-s", str(screen), "-t", str(threads), "-o", out_tsv]
    if os.path.exists(out_tsv):
        os.remove(out_tsv)
    t0 = time.perf_counter()
    subprocess.call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    dt = time.perf_counter() - t0
    if not os.path.isfile(out_tsv):
        raise RuntimeError("Had an issue running: " + " ".join(cmd))
    return {"seconds": dt, "version": version(exe), "command": " ".join(cmd), "rows": sum(1 for _ in open(out_tsv)) - 1}


def compare_tables(mine, theirs):
    """cell-by-cell differences of two 7-column tables (paths compared by base name; AF columns swapped when the
    two tables name a pair's genomes in opposite roles): dict with max / rms of |dANI| and |dAF| in percentage
    points, the number of cells that print identically, and the pairs only one table holds"""
    a, b = read_table(mine), read_table(theirs)
    d_ani, d_af, same_ani, same_af = [], [], 0, 0
    for k, g in b.items():
        if k not in a:
            continue
        m = a[k]
        afr, afq = (m[1], m[2]) if m[3] == g[3] else (m[2], m[1])
        d_ani.append(m[0] - g[0])
        d_af += [afr - g[1], afq - g[2]]
        same_ani += m[0] == g[0]
        same_af += (afr == g[1]) + (afq == g[2])
    n = len(d_ani)

    def stats(v):
        if not v:
            return 0.0, 0.0
        return max(abs(x) for x in v), (sum(x * x for x in v) / len(v)) ** 0.5
    ma, ra = stats(d_ani)
    mf, rf = stats(d_af)
    return {"pairs": n, "only_mine": len(set(a) - set(b)), "only_theirs": len(set(b) - set(a)),
            "max_abs_dANI": ma, "rms_dANI": ra, "max_abs_dAF": mf, "rms_dAF": rf,
            "identical_ANI_cells": same_ani, "identical_AF_cells": same_af, "unit": "percentage points"}
