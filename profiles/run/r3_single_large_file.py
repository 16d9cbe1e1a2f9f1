"""one large genome (100 Mb, 2,000 records, 80 columns) through skder_amd_sketch_n50-like ingest: tiled parser, wavefront-per-file
parser, host reader -- the ingest line of SKDER_AMD_DEBUG, best of three calls each"""
import os, re, subprocess, sys, tempfile
import numpy as np
rng = np.random.RandomState(3)
alpha = np.frombuffer(b"ACGT", np.uint8)
tmp = tempfile.mkdtemp(prefix="skder_amd_large_")
p = os.path.join(tmp, "big.fa")
with open(p, "wb") as f:
    for i in range(2000):
        body = alpha[rng.randint(0, 4, 50000)]
        f.write(b">contig%d\n" % i)
        f.write(np.concatenate([body.reshape(-1, 80), np.full((625, 1), 10, np.uint8)], axis=1).tobytes())
lst = os.path.join(tmp, "l.txt")
open(lst, "w").write(p + "\n" + p + "\n")
code = ("import sys,os; sys.path.insert(0,os.getcwd()); import skder_amd\n"
        "for _ in range(4): skder_amd.runSkaniTriangle(%r, %r, '-s 80', 0.0, 'greedy', False, None, n50_file=%r)\n") % (lst, os.path.join(tmp, "o.tsv"), os.path.join(tmp, "n.tsv"))
for mode, env in (("tiled", {}), ("wavefront per file", {"SKDER_AMD_FASTA_WAVE": "1"}), ("host reader", {"SKDER_AMD_HOST_PARSE": "1"})):
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SKDER_AMD_DEBUG="1", **env), capture_output=True, text=True)
    ms = [float(m) for m in re.findall(r"ingest of 2 files: ([0-9.]+) ms", r.stderr)]
    print(mode, "ingest of 2 x 101 MB:", min(ms[1:]) if len(ms) > 1 else r.stderr[-300:], "ms")
