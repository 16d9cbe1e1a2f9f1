"""round 4: BASELINE.json configs[4] AT ITS SIZE on ONE MI355X: N (50,000) synthetic genomes of 1 - 8 Mb (N/100 species), `--min-af 50`.
The sketch stage is STREAMED -- a batch of genomes is generated on the device, sketched and dropped, so the 225 GB of bases are never
resident --, everything else stays in HBM: raw sketches, index, marker table, work buffers, edge list.  Then the database calls of the
driver: skder_amd_db_triangle (screen, chaining, skani's row order in place, the parallel TSV writer) and the native selection.
Prints one JSON object with the phase times and the HBM budget (hipMemGetInfo after every phase)."""
import json, os, sys, tempfile, time
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
from skder_amd import engine, synth, selection
from skder_amd.skder import Database
import bench

N = int(os.environ.get("N", "50000"))
STEP = int(os.environ.get("STEP", "1000"))
ctx = engine.Context(0)
hbm = {}
def mem(tag):
    free, total = torch.cuda.mem_get_info()
    hbm[tag] = {"used_GB": round((total - free) / 1e9, 2), "total_GB": round(total / 1e9, 1)}
t_all = time.perf_counter()
t0 = time.perf_counter()
rec = synth.make_recipe(N, len_range=(1_000_000, 8_000_000))
t_recipe = time.perf_counter() - t0
total = sum(rec.total_len(g) for g in range(N))
sk = engine.Sketches(ctx)
sk.reserve(total // 120, total // 900)
mem("start")
t0 = time.perf_counter()
peak_batch = 0
for b0 in range(0, N, STEP):
    gs = range(b0, min(b0 + STEP, N))
    layout = engine.BatchLayout([rec.rec_lens[g] for g in gs])
    peak_batch = max(peak_batch, layout.total_bytes)
    d = torch.empty(layout.total_bytes, dtype=torch.uint8, device="cuda")
    ctx.synth_fill(d.data_ptr(), layout, rec.lineage[gs.start:gs.stop], rec.params[gs.start:gs.stop])
    sk.sketch_batch(d.data_ptr(), layout)
    del d
torch.cuda.synchronize()
t_sketch = time.perf_counter() - t0
torch.cuda.empty_cache()
mem("raw_sketches")
paths = ["/mixed/species%03d/g%05d.fasta" % (int(rec.species[g]), g) for g in range(N)]
n50 = [bench.n50_of_lengths(rec.rec_lens[g]) for g in range(N)]
t0 = time.perf_counter()
db = Database.from_sketches(sk, paths, n50, device=0)
sk.close()
t_db = time.perf_counter() - t0
mem("database_indexed")
tmp = tempfile.mkdtemp(prefix="skder_amd_scale_")
out_tsv = os.path.join(tmp, "Skani_Triangle_Edge_Output.txt")
t0 = time.perf_counter()
rows = db.triangle(50.0, 80.0, out_tsv=out_tsv)
t_tri = time.perf_counter() - t0
mem("after_triangle")
t0 = time.perf_counter()
rows2 = db.triangle(50.0, 80.0)
t_tri2 = time.perf_counter() - t0
table_bytes = os.path.getsize(out_tsv)
t0 = time.perf_counter()
reps = selection.native_greedy(rows, paths, n50, 99.5, 50.0)
t_sel = time.perf_counter() - t0
db.close()
per = N // max(N // 100, 1)
print(json.dumps({"workload": "%d synthetic genomes x 1.0-8.0 Mb (%d species x 10 strains x 10 isolates), triangle --min-af 50, screen 80, ONE MI355X; bases streamed through the sketch stage in batches of %d genomes" % (N, N // 100, STEP),
                  "genomes": N, "bases": int(total), "pairs": N * (N - 1) // 2, "rows": int(len(rows)), "rows_identical_second_call": bool(np.array_equal(rows, rows2)),
                  "seconds": {"recipe_host": t_recipe, "generate_and_sketch_streamed": t_sketch, "database_from_sketches_incl_index": t_db,
                              "db_triangle_with_table_on_disk": t_tri, "db_triangle_rows_in_memory": t_tri2, "native_greedy_selection": t_sel,
                              "all": time.perf_counter() - t_all},
                  "pairs_per_s_sketches_to_rows": N * (N - 1) / 2 / t_tri2, "table_bytes": table_bytes, "representatives": len(reps),
                  "hbm": hbm, "peak_input_batch_GB": round(peak_batch / 1e9, 2),
                  "command": "N=%d STEP=%d python profiles/run/r4_scale_50000.py" % (N, STEP)}))
import shutil; shutil.rmtree(tmp, ignore_errors=True)
