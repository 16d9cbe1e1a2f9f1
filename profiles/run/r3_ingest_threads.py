"""ingest scaling with host threads: the drop-in on 1024 sample files (plain, then gzip) for SKDER_AMD_IO_THREADS = 8 .. 256"""
import ctypes as C, json, os, subprocess, sys, time
sys.path.insert(0, os.getcwd())
import bench, torch
from skder_amd import _lib, engine, synth
ctx = engine.Context(0)
n = int(os.environ.get("N", "1024"))
recipe = synth.make_recipe(n, genome_len=3_000_000)
layout = engine.BatchLayout(recipe.rec_lens)
d = torch.empty(layout.total_bytes, dtype=torch.uint8, device="cuda")
ctx.synth_fill(d.data_ptr(), layout, recipe.lineage, recipe.params)
tmp, paths, nbytes = bench.write_sample_files([(layout, d)], n)
del d
gz, _ = bench.gzip_sample_files(paths)
res = {}
for kind, ps in (("plain", paths), ("gz", gz)):
    for th in (256, 128, 64, 32, 16, 8):
        env = dict(os.environ, SKDER_AMD_IO_THREADS=str(th), SKDER_AMD_DEBUG="1")
        code = ("import sys,os,ctypes as C,time; sys.path.insert(0,os.getcwd()); from skder_amd import _lib; import bench\n"
                "ps=open(%r).read().split()\n"
                "bench.end_to_end_sample(%r, ps[:8], 1, 0)\n"
                "bench.end_to_end_sample(%r, ps, 1, 0)\n"
                "r=bench.end_to_end_sample(%r, ps, %d, 0); print('RESULT', r['seconds'])\n") % (os.path.join(tmp, kind + ".lst"), tmp, tmp, tmp, nbytes)
        open(os.path.join(tmp, kind + ".lst"), "w").write("\n".join(ps))
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        ing = [l for l in p.stderr.splitlines() if "ingest of %d files" % n in l]
        sec = [l for l in p.stdout.splitlines() if l.startswith("RESULT")]
        res["%s_%d" % (kind, th)] = {"seconds": float(sec[-1].split()[1]) if sec else None, "ingest": ing[-1] if ing else p.stderr[-300:]}
        print(kind, th, res["%s_%d" % (kind, th)], flush=True)
print(json.dumps({"fasta_bytes": nbytes, "runs": res}))
