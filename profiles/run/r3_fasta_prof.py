"""the drop-in on 256 sample files with the FASTA parse on the device, for rocprofv3 --kernel-trace --stats"""
import os, sys
sys.path.insert(0, os.getcwd())
import bench, torch
from skder_amd import engine, synth
ctx = engine.Context(0)
n = int(os.environ.get("N", "256"))
recipe = synth.make_recipe(n, genome_len=3_000_000)
layout = engine.BatchLayout(recipe.rec_lens)
d = torch.empty(layout.total_bytes, dtype=torch.uint8, device="cuda")
ctx.synth_fill(d.data_ptr(), layout, recipe.lineage, recipe.params)
tmp, paths, nbytes = bench.write_sample_files([(layout, d)], n)
del d
for _ in range(3):
    r = bench.end_to_end_sample(tmp, paths, nbytes, 0)
print(r)
