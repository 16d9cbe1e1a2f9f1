"""ingest batch size: the drop-in on 1024 plain FASTA files for SKDER_AMD_IO_BATCH_MB = 64 .. 512, first call in the process and best of three later ones"""
import json, os, subprocess, sys
sys.path.insert(0, os.getcwd())
import bench, torch
from skder_amd import engine, synth
ctx = engine.Context(0)
n = 1024
recipe = synth.make_recipe(n, genome_len=3_000_000)
layout = engine.BatchLayout(recipe.rec_lens)
d = torch.empty(layout.total_bytes, dtype=torch.uint8, device="cuda")
ctx.synth_fill(d.data_ptr(), layout, recipe.lineage, recipe.params)
tmp, paths, nbytes = bench.write_sample_files([(layout, d)], n)
del d
open(os.path.join(tmp, "p.lst"), "w").write("\n".join(paths))
for mb in (64, 128, 256, 512):
    code = ("import sys,os; sys.path.insert(0,os.getcwd()); import bench\n"
            "ps=open(%r).read().split()\n"
            "bench.end_to_end_sample(%r, ps[:8], 1, 0)\n"
            "c=bench.end_to_end_sample(%r, ps, 1, 0)['seconds']\n"
            "w=min(bench.end_to_end_sample(%r, ps, 1, 0)['seconds'] for _ in range(3))\n"
            "print('RESULT', c, w)\n") % (os.path.join(tmp, "p.lst"), tmp, tmp, tmp)
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SKDER_AMD_IO_BATCH_MB=str(mb)), capture_output=True, text=True)
    print(mb, [l for l in p.stdout.splitlines() if l.startswith("RESULT")] or p.stderr[-300:], flush=True)
