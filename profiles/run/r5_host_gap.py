"""host time of the calls between two steps of the headline (GPU idle in between): which of them is the 1.2 ms gap?"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from skder_amd import engine, synth
ctx = engine.Context(0)
N = 5000
recipe = synth.make_recipe(N, genome_len=3_000_000)
batches = []
for b0 in range(0, N, 2500):
    gs = range(b0, b0 + 2500)
    layout = engine.BatchLayout([recipe.rec_lens[g] for g in gs])
    d = torch.empty(layout.total_bytes, dtype=torch.uint8, device="cuda")
    ctx.synth_fill(d.data_ptr(), layout, recipe.lineage[gs.start:gs.stop], recipe.params[gs.start:gs.stop])
    batches.append((layout, d))
total_bases = sum(l.total_bases for l, _ in batches)
torch.cuda.synchronize()
acc = {}
def T(name, f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    acc.setdefault(name, []).append(((t1 - t0) * 1e3, (t2 - t0) * 1e3)); return r
for it in range(5):
    sk = T("Sketches()", lambda: engine.Sketches(ctx))
    T("reserve", lambda: sk.reserve(total_bases // 120, total_bases // 900))
    for i, (layout, d) in enumerate(batches):
        T("sketch_batch[%d]" % i, lambda: sk.sketch_batch(d.data_ptr(), layout))
    edges = T("triangle_rows", lambda: sk.triangle_rows(0, 1, 80.0, copy=False))
    T("timing+counters", lambda: (ctx.timing(), ctx.index_ms(), ctx.runs_ms(), ctx.counters()))
    T("close", lambda: sk.close())
for k, v in acc.items():
    v = v[2:]
    print("%-18s returns after %.3f ms, GPU done after %.3f ms" % (k, np.mean([a for a, b in v]), np.mean([b for a, b in v])))
