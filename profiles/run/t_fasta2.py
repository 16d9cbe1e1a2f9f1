import os, sys, ctypes as C, tempfile, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle")); sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.path.join(os.getcwd(), "tests", "tools"))
os.environ["SKDER_AMD_DEBUG_FASTA"] = "1"
sys.argv = ["x", "0", "0"]
import fuzz_dropin as D
import fuzz_repeats as F
from skder_amd import _lib
import gzip
for seed in (8000000, 8000011):
    rng = np.random.RandomState(seed)
    anc = F.ancestor(rng); n = rng.randint(3, 7)
    gl = [F.descend(rng, anc, i) for i in range(n)]
    tmp = tempfile.mkdtemp(prefix="fz4_")
    paths = []
    for i, (seq, lens) in enumerate(gl):
        pth = os.path.join(tmp, "g%d.fa%s" % (i, ".gz" if rng.rand() < 0.3 else ""))
        D.write_fasta(rng, pth, seq, lens); paths.append(pth)
        data = gzip.open(pth, "rb").read() if pth.endswith(".gz") else open(pth, "rb").read()
        print(os.path.basename(pth), len(data), repr(data[:70]), "records", len(lens), list(lens[:4]))
    listing = os.path.join(tmp, "list.txt"); open(listing, "w").write("".join(q + "\n" for q in paths))
    err = C.create_string_buffer(2048)
    a = os.path.join(tmp, "gpu.tsv")
    rc = _lib.lib().skder_amd_triangle(listing.encode(), 0.0, 0.0, 0, a.encode(), err, 2048)
    print(rc, err.value)
