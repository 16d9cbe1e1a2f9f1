"""round 5: the seed index build alone (index_genome_lds_kernel + chunk tables) on the headline's sketches: N genomes x 3 Mb, every kernel
alone on the chip (nothing beside it), milliseconds per build; rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE around the same script gives the traffic"""
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from skder_amd import engine, multigpu, synth
N = int(os.environ.get("N", "5000"))
REPS = int(os.environ.get("REPS", "5"))
ctx = engine.Context(0)
rec = synth.make_recipe(N, genome_len=3_000_000)
sk = engine.Sketches(ctx)
for b0 in range(0, N, 1250):
    gs = range(b0, min(b0 + 1250, N))
    layout = engine.BatchLayout([rec.rec_lens[g] for g in gs])
    d = torch.empty(layout.total_bytes, dtype=torch.uint8, device="cuda")
    ctx.synth_fill(d.data_ptr(), layout, rec.lineage[gs.start:gs.stop], rec.params[gs.start:gs.stop])
    sk.sketch_batch(d.data_ptr(), layout)
    del d
torch.cuda.synchronize()
raw = multigpu.raw_from_sketches(sk, copy=True)
ms = []
for _ in range(REPS):
    s2 = multigpu.sketches_from_raw(ctx, raw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s2.index()
    torch.cuda.synchronize()
    ms.append((time.perf_counter() - t0) * 1e3)
    s2.close()
print(json.dumps({"genomes": N, "seeds": int(raw["seed_kmer"].numel()), "index_wall_ms": [round(x, 3) for x in ms], "index_kernel_ms_last": ctx.index_ms()}))
