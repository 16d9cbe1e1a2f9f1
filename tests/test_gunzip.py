"""The gzip decoder of the ingest (skder_amd/csrc/gunzip.cpp, host code) against zlib, on the CPU, under AddressSanitizer +
UBSan (tests/gunzip_harness.cpp): every block type (stored, fixed, dynamic; Huffman-only, RLE, 15-bit codes), every header
option (FEXTRA, FNAME, FCOMMENT, FHCRC), several members, empty members, flushes, trailing bytes; output buffers that are too
small; a few thousand damaged streams, where the decoder must fail exactly where zlib fails and never touch memory outside
its two buffers; the CRC-32 (carry-less multiplication / tables) against zlib's on random lengths and alignments."""
import os
import random
import struct
import subprocess
import zlib

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "skder_amd", "csrc")


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    d = tmp_path_factory.mktemp("gunzip")
    exe = str(d / "gunzip_harness")
    subprocess.check_call(["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-std=c++17",
                           os.path.join(ROOT, "tests", "gunzip_harness.cpp"), os.path.join(CSRC, "gunzip.cpp"), "-o", exe, "-lz"])
    return exe, d


def _run(exe, *args):
    out = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert "ERROR" not in out.stderr and "runtime error" not in out.stderr, out.stderr[-3000:]
    return out


def _member(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, flg=0, extra=b"", name=b"", comment=b"", memlevel=8):
    hdr = b"\x1f\x8b\x08" + bytes([flg]) + b"\0\0\0\0" + b"\0\xff"
    if flg & 4:
        hdr += struct.pack("<H", len(extra)) + extra
    if flg & 8:
        hdr += name + b"\0"
    if flg & 16:
        hdr += comment + b"\0"
    if flg & 2:
        hdr += struct.pack("<H", zlib.crc32(hdr) & 0xFFFF)
    c = zlib.compressobj(level, zlib.DEFLATED, -15, memlevel, strategy)
    return hdr + c.compress(data) + c.flush() + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data) & 0xFFFFFFFF)


def _corpus():
    rng = random.Random(5)
    s = "".join(rng.choice("ACGT") for _ in range(200000))
    s = s[:70000] + s[:70000][::-1][:30000] + s[:100000]
    dna = (">seq1 test\n" + "\n".join(s[i:i + 70] for i in range(0, len(s), 70)) + "\n").encode()
    texts = {"dna": dna, "small": b">a\nACGT\n", "empty": b"", "bin": bytes(rng.getrandbits(8) for _ in range(70000)),
             "text": b"the quick brown fox jumps over the lazy dog. " * 3000, "zeros": bytes(200000), "one": b"A"}
    files = {}
    for tn, t in texts.items():
        for lv in (0, 1, 6, 9):
            files["%s_l%d" % (tn, lv)] = _member(t, lv)
        files[tn + "_fixed"] = _member(t, 6, zlib.Z_FIXED)
        files[tn + "_huff"] = _member(t, 6, zlib.Z_HUFFMAN_ONLY)
        files[tn + "_rle"] = _member(t, 6, zlib.Z_RLE)
        files[tn + "_mem1"] = _member(t, 9, memlevel=1)
    files["hdr_all"] = _member(dna, 1, flg=4 | 8 | 16 | 2, extra=b"BC\x02\x00\x10\x00", name=b"file.fa", comment=b"hello")
    files["hdr_name"] = _member(texts["text"], 6, flg=8, name=b"x" * 300)
    files["multi"] = _member(dna[:100000], 1) + _member(dna[100000:], 9) + _member(b"", 6) + _member(texts["small"], 6)
    files["trailing"] = _member(dna, 6) + b"\0\0\0garbage that is no member"
    files["bgzf_like"] = b"".join(_member(dna[i:i + 65000], 6, flg=4, extra=b"BC\x02\x00\xff\xff") for i in range(0, len(dna), 65000))
    c = zlib.compressobj(6, zlib.DEFLATED, 31)
    files["flushed"] = (c.compress(dna[:50000]) + c.flush(zlib.Z_SYNC_FLUSH) + c.compress(dna[50000:]) + c.flush(zlib.Z_FULL_FLUSH) + c.flush())
    # a skewed alphabet that forces 15-bit codes (second-level tables)
    sk = bytearray()
    for i in range(40):
        sk += bytes([i]) * (2 ** min(i, 20) if i < 21 else 1)
    lst = list(bytes(sk[:1500000]))
    rng.shuffle(lst)
    files["skewed"] = _member(bytes(lst), 6, zlib.Z_HUFFMAN_ONLY)
    return files, texts


def test_every_block_type_and_header_option_equals_zlib(harness):
    exe, d = harness
    files, _ = _corpus()
    for name, blob in files.items():
        f = d / (name + ".gz")
        f.write_bytes(blob)
        out = _run(exe, "check", f)
        assert out.returncode == 0 and out.stdout.startswith("same"), (name, out.stdout, out.stderr[-500:])


def test_the_reference_s_own_genome_files_decode_like_zlib(harness):
    """the 34 real .fasta.gz assemblies of the reference's test run (tests/golden/genomes): NCBI's gzip, 3.5 x, 2.6 matches per literal"""
    exe, _ = harness
    gdir = os.path.join(ROOT, "tests", "golden", "genomes")
    files = sorted(f for f in os.listdir(gdir) if f.endswith(".gz"))
    assert len(files) >= 30
    for f in files:
        out = _run(exe, "check", os.path.join(gdir, f))
        assert out.returncode == 0 and out.stdout.startswith("same"), (f, out.stdout, out.stderr[-300:])


def test_output_buffer_too_small_is_reported_not_overrun(harness):
    exe, d = harness
    files, texts = _corpus()
    for name, text in (("dna_l6", texts["dna"]), ("text_l6", texts["text"]), ("zeros_l6", texts["zeros"]), ("dna_l0", texts["dna"]), ("multi", None)):
        f = d / (name + ".gz")
        f.write_bytes(files[name])
        n = len(text) if text is not None else len(texts["dna"]) + len(texts["small"])
        for cap in (0, 1, 100, n // 2, n - 300, n - 1):
            out = _run(exe, "check", f, cap)
            assert out.returncode == 0 and out.stdout.startswith("full"), (name, cap, out.stdout)
        out = _run(exe, "check", f, n)
        assert out.returncode == 0 and out.stdout.startswith("same"), (name, out.stdout)


def test_damaged_streams_fail_like_zlib_and_stay_inside_the_buffers(harness):
    exe, d = harness
    files, _ = _corpus()
    for k, name in enumerate(("dna_l1", "dna_l6", "multi", "bin_l6", "text_l6", "zeros_l6", "skewed", "flushed", "hdr_all", "small_l6",
                              "empty_l6", "bgzf_like", "trailing", "dna_fixed", "dna_l0", "text_rle")):
        f = d / (name + ".gz")
        f.write_bytes(files[name])
        out = _run(exe, "fuzz", f, 100 + k, 250)
        assert out.returncode == 0 and "0 mismatches" in out.stdout, (name, out.stdout[-600:])


def test_crc32_equals_zlib(harness):
    exe, _ = harness
    out = _run(exe, "crc", 3, 4000)
    assert out.returncode == 0 and "0 mismatches" in out.stdout, out.stdout
