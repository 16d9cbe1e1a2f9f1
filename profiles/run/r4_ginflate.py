"""round 4: the device-side DEFLATE decoder measured -- N_STREAMS gzip members (the reference's 34 real .fasta.gz genomes, repeated; level of
the files as they are) resident in HBM, one wavefront per stream: kernel milliseconds, GB/s of text, per-stream rate; beside it the host
decoders on this box's CPUs (zlib and the ingest's own gunzip.cpp through the file-based drop-in are measured by bench.py's end_to_end_gz)"""
import ctypes as C, gzip, json, os, sys, time, zlib
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from skder_amd import engine, _lib
from test_ginflate import JOB, RES, gz_payload
N = int(os.environ.get("N_STREAMS", "5000"))
gold = os.path.join("tests", "golden", "genomes")
names = sorted(os.listdir(gold))
blobs = [open(os.path.join(gold, n), "rb").read() for n in names]
meta = [gz_payload(b) for b in blobs]
streams = [np.frombuffer(b[m[0]:-8], np.uint8) for b, m in zip(blobs, meta)]
ctx = engine.Context(0)
jobs = np.zeros(N, JOB)
ioff = ooff = 0
for k in range(N):
    s, m = streams[k % 34], meta[k % 34]
    ioff = (ioff + 7) & ~7; ooff = (ooff + 31) & ~31
    jobs[k] = (ioff, len(s), 0, ooff, m[2]); ioff += len(s); ooff += m[2]
host = np.zeros(ioff + 64, np.uint8)
for k in range(N):
    o = int(jobs[k]["in_off"]); host[o:o + int(jobs[k]["in_len"])] = streams[k % 34]
t0 = time.perf_counter(); d_in = torch.from_numpy(host).pin_memory().cuda(non_blocking=False); torch.cuda.synchronize(); t_h2d = time.perf_counter() - t0
d_out = torch.empty(ooff + 64, dtype=torch.uint8, device="cuda")
res = np.zeros(N, RES); ms = (C.c_float * 2)()
best = None
for rep in range(3):
    t0 = time.perf_counter()
    rc = _lib.lib().skder_amd_inflate_device(ctx.h, d_in.data_ptr(), jobs.ctypes.data, N, d_out.data_ptr(), res.ctypes.data, ms)
    wall = time.perf_counter() - t0
    assert rc == 0
    if best is None or ms[0] < best[0]: best = (ms[0], ms[1], wall)
ok = int((res["status"] == 0).sum())
crc_ok = int(sum(int(res[k]["crc"]) == meta[k % 34][1] for k in range(N)))
text = int(res["out_len"].sum())
# one host thread of this box on the same streams (zlib), for the per-stream comparison
t0 = time.perf_counter()
for k in range(34): zlib.decompress(bytes(streams[k]), -15)
t_zlib = time.perf_counter() - t0
print(json.dumps({"streams": N, "compressed_bytes": int(ioff), "text_bytes": text, "ratio": text / ioff, "status_ok": ok, "crc_ok": crc_ok,
                  "inflate_kernel_ms": best[0], "crc_kernel_ms": best[1], "call_wall_ms": best[2] * 1e3, "h2d_compressed_ms": t_h2d * 1e3,
                  "text_GB_per_s_kernel": text / best[0] / 1e6, "text_GB_per_s_incl_crc": text / (best[0] + best[1]) / 1e6,
                  "per_stream_MB_per_s_at_full_occupancy": text / best[0] / 1e3 / min(N, 1024),
                  "zlib_one_host_thread_MB_per_s": sum(m[2] for m in meta) / t_zlib / 1e6,
                  "what": "one wavefront per stream, 4 per CU (40.5 KB of LDS each): the 34 real C. granulosum .fasta.gz of the reference's test run, repeated"}))
