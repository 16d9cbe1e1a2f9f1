import ctypes as C, gzip, os, sys, zlib
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np, torch
from skder_amd import engine, _lib
from test_ginflate import JOB, RES, raw_deflate
ctx = engine.Context(0)
real = gzip.decompress(open("tests/golden/genomes/Cutibacterium_granulosum_GCA_900186975.1.fasta.gz","rb").read())
rng = np.random.RandomState(1)
dna = bytes(np.frombuffer(b"ACGT", np.uint8)[rng.randint(0,4,2100000)])
synth = b">r1\n" + b"\n".join(dna[i:i+60] for i in range(0,len(dna),60)) + b"\n"
def run(name, data, level):
    s = raw_deflate(data, level)
    jobs = np.zeros(1, JOB); jobs[0] = (0, len(s), 0, 64, len(data))
    host = np.zeros(len(s)+64, np.uint8); host[:len(s)] = np.frombuffer(s, np.uint8)
    d_in = torch.from_numpy(host).cuda(); d_out = torch.empty(len(data)+256, dtype=torch.uint8, device="cuda")
    res = np.zeros(1, RES); ms = (C.c_float*2)()
    for _ in range(2):
        assert _lib.lib().skder_amd_inflate_device(ctx.h, d_in.data_ptr(), jobs.ctypes.data, 1, d_out.data_ptr(), res.ctypes.data, ms) == 0
    # symbols: count via zlib? approximate from compressed size
    print("%-14s level %d: ratio %.2f, status %d, %.1f ms -> %.1f MB/s per stream" % (name, level, len(data)/len(s), res[0]["status"], ms[0], len(data)/ms[0]/1e3))
for lv in (1, 6, 9):
    run("real genome", real, lv)
    run("random DNA", synth, lv)
