"""round 5: would an INCREMENTAL triangle pay?  The sketch of the second half of the genomes (VALU-bound) beside the triangle of the first half
(index, screen, join, run extraction, chaining): two contexts on one GPU, two host threads (ctypes releases the GIL), each looped REPS times --
one after the other, then side by side.  If side by side takes about as long as the two sums, the chip has nothing to overlap."""
import json, os, sys, threading, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from skder_amd import engine, multigpu, synth
N, REPS = 5000, int(os.environ.get("REPS", "6"))
rec = synth.make_recipe(N, genome_len=3_000_000)
ctxA, ctxB = engine.Context(0), engine.Context(0)
half = N // 2
def bases(ctx, lo, hi):
    lay = engine.BatchLayout([rec.rec_lens[g] for g in range(lo, hi)])
    d = torch.empty(lay.total_bytes, dtype=torch.uint8, device="cuda")
    ctx.synth_fill(d.data_ptr(), lay, rec.lineage[lo:hi], rec.params[lo:hi])
    return lay, d
lay1, d1 = bases(ctxB, 0, half)
lay2, d2 = bases(ctxA, half, N)
torch.cuda.synchronize()
def sketch_second():
    for _ in range(REPS):
        s = engine.Sketches(ctxA); s.sketch_batch(d2.data_ptr(), lay2); s.close()
def triangle_first():
    for _ in range(REPS):
        s = engine.Sketches(ctxB); s.sketch_batch(d1.data_ptr(), lay1); s.triangle_rows(0, 1, 80.0, copy=False); s.close()
def first_sketch_only():
    for _ in range(REPS):
        s = engine.Sketches(ctxB); s.sketch_batch(d1.data_ptr(), lay1); s.close()
def timed(*fns):
    th = [threading.Thread(target=f) for f in fns]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / REPS * 1e3
for f in (sketch_second, triangle_first):
    timed(f)                                  # warm-up
out = {"sketch_second_half_ms": timed(sketch_second), "sketch_plus_triangle_first_half_ms": timed(triangle_first), "sketch_first_half_ms": timed(first_sketch_only)}
out["side_by_side_ms"] = timed(sketch_second, triangle_first)
out["one_after_the_other_ms"] = out["sketch_second_half_ms"] + out["sketch_plus_triangle_first_half_ms"]
out["triangle_first_half_alone_ms"] = out["sketch_plus_triangle_first_half_ms"] - out["sketch_first_half_ms"]
print(json.dumps(out))
