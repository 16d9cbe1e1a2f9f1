"""GPU tests of the device-side DEFLATE decoder (skder_amd/csrc/ginflate.hip): byte-equal with zlib on the reference's 34 .fasta.gz genomes and
on streams that exercise every block type (stored, fixed, dynamic with second-level tables, overlapping matches, many blocks); the CRC-32 of
the second kernel equals zlib's; damaged streams are rejected (status, CRC or length) and nothing is written outside a stream's region."""
import ctypes as C
import gzip
import os
import zlib

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

JOB = np.dtype([("in_off", "<u8"), ("in_len", "<u4"), ("pad", "<u4"), ("out_off", "<u8"), ("out_cap", "<u8")])
RES = np.dtype([("status", "<u4"), ("crc", "<u4"), ("in_used", "<u4"), ("pad", "<u4"), ("out_len", "<u8")])


@pytest.fixture(scope="module")
def gpu():
    import torch
    from skder_amd import engine
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    ctx = engine.Context(0)
    yield engine, ctx, torch
    ctx.close()


def gz_payload(blob: bytes):
    """(offset of the DEFLATE stream, stored CRC-32, stored ISIZE) of a one-member gzip file"""
    assert blob[:2] == b"\x1f\x8b" and blob[2] == 8
    flg, p = blob[3], 10
    if flg & 4:
        p += 2 + (blob[p] | blob[p + 1] << 8)
    for bit in (8, 16):
        if flg & bit:
            p = blob.index(b"\0", p) + 1
    if flg & 2:
        p += 2
    return p, int.from_bytes(blob[-8:-4], "little"), int.from_bytes(blob[-4:], "little")


def raw_deflate(data: bytes, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, mem=8):
    c = zlib.compressobj(level, zlib.DEFLATED, -15, mem, strategy)
    return c.compress(data) + c.flush()


def inflate_all(gpu, streams, caps, guard=64):
    """streams: list of raw DEFLATE byte strings; caps: output capacity of each -> (results, list of output arrays, guards intact)"""
    from skder_amd import _lib
    engine, ctx, torch = gpu
    jobs = np.zeros(len(streams), JOB)
    ioff = ooff = 0
    for k, (s, cap) in enumerate(zip(streams, caps)):
        ioff = (ioff + 7) & ~7
        ooff = (ooff + guard + 31) & ~31
        jobs[k] = (ioff, len(s), 0, ooff, cap)
        ioff += len(s)
        ooff += cap
    host_in = np.zeros(ioff + 64, np.uint8)
    for k, s in enumerate(streams):
        host_in[int(jobs[k]["in_off"]):int(jobs[k]["in_off"]) + len(s)] = np.frombuffer(s, np.uint8)
    d_in = torch.from_numpy(host_in).cuda()
    d_out = torch.full((ooff + guard + 64,), 0xEE, dtype=torch.uint8, device="cuda")
    res = np.zeros(len(streams), RES)
    ms = (C.c_float * 2)()
    rc = _lib.lib().skder_amd_inflate_device(ctx.h, d_in.data_ptr(), jobs.ctypes.data, len(streams), d_out.data_ptr(), res.ctypes.data, ms)
    assert rc == 0, ctx.last_error() if hasattr(ctx, "last_error") else rc
    out = d_out.cpu().numpy()
    outs, intact = [], True
    used = np.zeros(len(out), bool)
    for k in range(len(streams)):
        o, cap = int(jobs[k]["out_off"]), int(jobs[k]["out_cap"])
        outs.append(out[o:o + min(int(res[k]["out_len"]), cap)])
        used[o:o + cap] = True
    intact = bool((out[~used] == 0xEE).all())
    return res, outs, intact, (ms[0], ms[1])


def test_reference_genomes_byte_equal_with_zlib(gpu):
    names = sorted(os.listdir(os.path.join(GOLDEN, "genomes")))
    blobs = [open(os.path.join(GOLDEN, "genomes", n), "rb").read() for n in names]
    texts = [gzip.decompress(b) for b in blobs]
    meta = [gz_payload(b) for b in blobs]
    streams = [b[m[0]:-8] for b, m in zip(blobs, meta)]
    res, outs, intact, ms = inflate_all(gpu, streams, [m[2] for m in meta])
    assert intact
    for k, t in enumerate(texts):
        assert res[k]["status"] == 0, (names[k], res[k])
        assert int(res[k]["out_len"]) == len(t) == meta[k][2] and int(res[k]["in_used"]) == len(streams[k])
        assert bytes(outs[k]) == t, names[k]
        assert int(res[k]["crc"]) == meta[k][1] == zlib.crc32(t)
    total = sum(len(t) for t in texts)
    print("34 genomes, %.1f MB of text: inflate %.2f ms (%.1f GB/s with 34 streams in flight), crc %.2f ms" % (total / 1e6, ms[0], total / ms[0] / 1e6, ms[1]))


def test_every_block_type(gpu):
    rng = np.random.RandomState(5)
    dna = bytes(np.frombuffer(b"ACGT", np.uint8)[rng.randint(0, 4, 300000)])
    fasta = b">r1\n" + b"\n".join(dna[i:i + 60] for i in range(0, 200000, 60)) + b"\n"
    rnd = bytes(rng.randint(0, 256, 70000).astype(np.uint8))
    skew = bytes(np.minimum(rng.geometric(0.02, 120000), 255).astype(np.uint8))          # long codes: second-level tables
    cases = {
        "empty": raw_deflate(b""),
        "one_byte": raw_deflate(b"A"),
        "stored": raw_deflate(fasta[:70000], 0),
        "stored_random": raw_deflate(rnd, 6),                                               # zlib falls back to stored blocks
        "fixed": raw_deflate(fasta[:5000], 6, zlib.Z_FIXED),
        "dynamic_1": raw_deflate(fasta, 1),
        "dynamic_9": raw_deflate(fasta, 9),
        "huffman_only": raw_deflate(skew, 6, zlib.Z_HUFFMAN_ONLY),
        "rle": raw_deflate(b"N" * 100000 + fasta[:1000] + b"AC" * 5000 + b"ACG" * 7000, 6, zlib.Z_RLE),
        "overlaps": raw_deflate(b"A" * 300 + b"AC" * 400 + b"ACGTT" * 300 + dna[:100] * 50, 9),
        "many_blocks": raw_deflate(fasta, 6, zlib.Z_DEFAULT_STRATEGY, 1),                   # memLevel 1: a block every 128 symbols
        "skewed": raw_deflate(skew, 9),
        "window_edge": raw_deflate(dna[:40000] + dna[:40000] + dna[7232:40000 + 7232], 9),
    }
    want = {k: zlib.decompress(v, -15) for k, v in cases.items()}
    keys = sorted(cases)
    res, outs, intact, _ = inflate_all(gpu, [cases[k] for k in keys], [len(want[k]) for k in keys])
    assert intact
    for i, k in enumerate(keys):
        assert res[i]["status"] == 0, (k, res[i])
        assert bytes(outs[i]) == want[k], k
        assert int(res[i]["crc"]) == zlib.crc32(want[k]), k
        assert int(res[i]["in_used"]) == len(cases[k]), k
    # an output region that is too small is an error, never an overrun
    res, outs, intact, _ = inflate_all(gpu, [cases["dynamic_1"], cases["stored"]], [len(want["dynamic_1"]) - 1, 1000])
    assert intact and res[0]["status"] == 4 and res[1]["status"] == 4


def test_damaged_streams_are_rejected(gpu):
    """4,000 damaged copies of real streams (a flipped byte, a cut, garbage appended inside the counted length): each must end with a
    status, or decode to something its CRC / length does not confirm; nothing outside the regions is written"""
    rng = np.random.RandomState(11)
    names = sorted(os.listdir(os.path.join(GOLDEN, "genomes")))[:4]
    good = []
    for n in names:
        b = open(os.path.join(GOLDEN, "genomes", n), "rb").read()
        m = gz_payload(b)
        text = gzip.decompress(b)[:200000]
        good.append((raw_deflate(text, 6), zlib.crc32(text), len(text)))
    streams, want = [], []
    for k in range(4000):
        s, crc, n = good[k % len(good)]
        s = bytearray(s)
        kind = k % 3
        if kind == 0:
            i = rng.randint(0, len(s))
            s[i] ^= 1 << rng.randint(0, 8)
        elif kind == 1:
            s = s[:rng.randint(1, len(s))]
        else:
            i = rng.randint(0, len(s) - 8)
            s[i:i + 8] = bytes(rng.randint(0, 256, 8).astype(np.uint8))
        streams.append(bytes(s))
        want.append((crc, n))
    res, outs, intact, _ = inflate_all(gpu, streams, [w[1] + 300 for w in want])
    assert intact
    accepted = 0
    for k in range(len(streams)):
        ok = res[k]["status"] == 0 and int(res[k]["out_len"]) == want[k][1] and int(res[k]["crc"]) == want[k][0]
        if ok:
            # only if zlib itself accepts the damaged stream with the same text (a flipped bit in padding, a cut behind the last block)
            try:
                d = zlib.decompressobj(-15)
                t = d.decompress(streams[k])
                assert d.eof and zlib.crc32(t) == want[k][0], k
            except zlib.error:
                raise AssertionError("stream %d accepted by the device, refused by zlib" % k)
            accepted += 1
    assert accepted < 200
