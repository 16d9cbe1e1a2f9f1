"""The FASTA parser of the ingest path under AddressSanitizer + UBSan (CPU build; GPU sanitizers are not
available on the pool).  Edge cases follow util.n50_calc (/root/reference/src/skDER/util.py:686-724) and the
reader the reference delegates to skani: CRLF files, blanks inside lines, text before the first header, empty
records, a last line without newline, thousands of short records, gzip input."""
import gzip
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "skder_amd", "csrc")


def _n50(lens):
    lens = sorted((l for l in lens if l), reverse=True)
    half, cum = int(sum(lens) / 2), 0
    for l in lens:
        cum += l
        if cum >= half:
            return l


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not available")
def test_parser_under_sanitizers(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    subprocess.check_call(["make", "-C", CSRC, "-j8"], stdout=subprocess.DEVNULL)
    # every object of the library except host_io.o, which the harness compiles from source under the sanitizers (list: csrc/Makefile)
    mk = open(os.path.join(CSRC, "Makefile")).read()
    srcs = re.search(r"^SRC = (.*)$", mk, re.M).group(1).split()
    objs = [os.path.join(CSRC, f[:-4] + ".o") for f in srcs if f != "host_io.hip"] + [os.path.join(CSRC, o) for o in ("gunzip.o", "select.o")]
    exe = str(tmp_path / "harness")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O1", "-g", "-Xarch_host", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",   # host code only
                           "-std=c++17", "-I" + CSRC, "-x", "hip", os.path.join(ROOT, "tests", "host_parser_harness.cpp"),
                           os.path.join(CSRC, "host_io.hip"), "-x", "none"] + objs + ["-o", exe, "-lz", "-lpthread"],
                          stderr=subprocess.DEVNULL)
    rng = np.random.RandomState(3)
    alpha = np.frombuffer(b"ACGT", np.uint8)
    seq = lambda n: bytes(alpha[rng.randint(0, 4, n)])
    wrap = lambda s, w, eol=b"\n": eol.join(s[i:i + w] for i in range(0, len(s), w)) + eol
    files, want = {}, {}
    files["plain.fa"] = b">r1 x\n" + wrap(seq(70001), 80) + b">short\n" + wrap(seq(300), 60) + b">r3\n" + wrap(seq(555), 70)
    want["plain.fa"] = (2, 70556, _n50([70001, 300, 555]), "r1 x")
    files["crlf.fa"] = b">r1\r\n" + wrap(seq(5000), 60, b"\r\n") + b">r2\r\n" + wrap(seq(499), 60, b"\r\n")
    want["crlf.fa"] = (1, 5000, _n50([5000, 499]), "r1")
    body = wrap(seq(3000), 50).replace(b"A", b"A ", 20)
    files["blanks.fa"] = b"ACGT\n>r1\n" + body + b"\n\n>e\n\n>r2\n  " + seq(700) + b"  \n"
    want["blanks.fa"] = (2, 3700, _n50([4, 3020, 700]), "r1")          # inner blanks count for N50 (len(line.strip()))
    files["noeol.fa"] = b">r1\n" + seq(1200)
    want["noeol.fa"] = (1, 1200, 1200, "r1")
    lens = [500 + i % 37 for i in range(3000)]
    files["many.fa"] = b"".join(b">c%d\n" % i + wrap(seq(l), 61) for i, l in enumerate(lens))
    want["many.fa"] = (3000, sum(lens), _n50(lens), "c0")
    for k, v in files.items():
        (tmp_path / k).write_bytes(v)
    for k in ("plain.fa", "many.fa"):
        with gzip.open(tmp_path / (k + ".gz"), "wb") as f:
            f.write(files[k])
        want[k + ".gz"] = want[k]
    # several gzip members in one file decode to the concatenation of their texts (zlib's gzread does that; bgzip writes files
    # of thousands of small members), member boundaries anywhere -- inside a line, inside a header; bytes behind the last
    # member that do not start another one are ignored
    text = files["many.fa"]
    cuts = [0, 7, 1000, 1003, len(text) // 2, len(text) - 5, len(text)]
    (tmp_path / "members.fa.gz").write_bytes(b"".join(gzip.compress(text[a:b], 1) for a, b in zip(cuts[:-1], cuts[1:])))
    want["members.fa.gz"] = want["many.fa"]
    (tmp_path / "blocks.fa.gz").write_bytes(b"".join(gzip.compress(files["plain.fa"][i:i + 997], 6) for i in range(0, len(files["plain.fa"]), 997)) + gzip.compress(b""))
    want["blocks.fa.gz"] = want["plain.fa"]
    (tmp_path / "tail.fa.gz").write_bytes(gzip.compress(files["crlf.fa"]) + b"\0" * 37 + b"not a member")
    want["tail.fa.gz"] = want["crlf.fa"]
    # a gzip stream cut short must be refused, not taken for a shorter genome -- in the first member or in a later one
    whole = (tmp_path / "plain.fa.gz").read_bytes()
    (tmp_path / "cut.fa.gz").write_bytes(whole[:len(whole) * 2 // 3])
    whole = (tmp_path / "members.fa.gz").read_bytes()
    (tmp_path / "cut2.fa.gz").write_bytes(whole[:len(whole) - 9])
    names = sorted(want) + ["cut.fa.gz", "cut2.fa.gz"]
    out = subprocess.run([exe] + names, cwd=tmp_path, capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"),
                         timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "ERROR" not in out.stderr and "runtime error" not in out.stderr, out.stderr[-3000:]
    lines = [l.split() for l in out.stdout.splitlines()]
    assert len(lines) == len(names)
    for c in ("cut.fa.gz", "cut2.fa.gz"):
        cut = [l for l in lines if l[0] == c]
        assert len(cut) == 1 and cut[0][1] == "ERRORED" and "gzip" in " ".join(cut[0]), cut
    for l in [l for l in lines if l[0] not in ("cut.fa.gz", "cut2.fa.gz")]:
        nrec, nbases, n50, first = want[l[0]]
        assert int(l[2]) == nrec and int(l[4]) == nbases and int(l[8]) == n50, l
        assert l[-2:] == ["same", "1"], l                    # own-memory and region layouts agree byte for byte
        assert ("'" + first + "'") in " ".join(l), l
