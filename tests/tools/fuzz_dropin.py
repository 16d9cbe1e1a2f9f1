"""randomized drop-in parity: FASTA files with random formatting -> skder_amd_triangle vs oracle.triangle (text equality)"""
import os, sys, time, gzip, ctypes as C, tempfile, shutil
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle")); sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.path.join(os.getcwd(), "tests", "tools"))
import numpy as np
import oracle_py as oracle
from skder_amd import _lib
import fuzz_repeats as F

def write_fasta(rng, path, seq, lens):
    eol = b"\r\n" if rng.rand() < 0.2 else b"\n"
    width = int(rng.choice([60, 70, 80, 1000, 10 ** 9, rng.randint(1, 200)]))
    out = []
    if rng.rand() < 0.1: out.append(eol)                      # leading blank line
    pos = 0
    for i, l in enumerate(lens):
        out.append(b">rec%d some description" % i + eol)
        body = bytes(seq[pos:pos + int(l)]); pos += int(l)
        for a in range(0, len(body), width):
            out.append(body[a:a + width] + eol)
            if rng.rand() < 0.002: out.append(eol)            # blank line inside a record
    data = b"".join(out)
    if rng.rand() < 0.1 and data.endswith(eol): data = data[:-len(eol)]   # no newline at the end
    if path.endswith(".gz"):
        with gzip.open(path, "wb") as f: f.write(data)
    else:
        open(path, "wb").write(data)

def main():
    p = oracle.default_params(); bad = 0; t0 = time.time()
    for seed in range(int(sys.argv[1]), int(sys.argv[2])):
        rng = np.random.RandomState(seed)
        anc = F.ancestor(rng); n = rng.randint(3, 7)
        gl = [F.descend(rng, anc, i) for i in range(n)]
        tmp = tempfile.mkdtemp(prefix="fz4_")
        try:
            paths = []
            for i, (seq, lens) in enumerate(gl):
                pth = os.path.join(tmp, "g%d.fa%s" % (i, ".gz" if rng.rand() < 0.3 else ""))
                write_fasta(rng, pth, seq, lens); paths.append(pth)
            listing = os.path.join(tmp, "list.txt"); open(listing, "w").write("".join(q + "\n" for q in paths))
            screen = 80.0 if rng.rand() < 0.5 else 0.0
            min_af = float(rng.choice([0.0, 15.0, 50.0]))
            a, b = os.path.join(tmp, "gpu.tsv"), os.path.join(tmp, "orc.tsv")
            err = C.create_string_buffer(2048)
            rc = _lib.lib().skder_amd_triangle(listing.encode(), min_af, screen, 0, a.encode(), err, 2048)
            assert rc == 0, err.value
            oracle.triangle(listing, min_af, screen, 4, b, p)
            if open(a).read() != open(b).read():
                bad += 1; print("seed", seed, "MISMATCH"); 
                la, lb = open(a).read().splitlines(), open(b).read().splitlines()
                for x, y in zip(la, lb):
                    if x != y: print("  gpu:", x); print("  orc:", y); break
                print("  lines", len(la), len(lb), flush=True)
            else:
                print("seed", seed, "ok", flush=True)
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    print("done", bad, "mismatches in", round(time.time() - t0, 1), "s")
main()
