"""round 4: the DEFAULT mode (greedy / dynamic: `skani triangle` + selection on the table) on N genomes of ONE species -- every pair passes the
screen and is chained.  The 34 real assemblies + device-generated descendants, sketches resident; db.triangle (--min-af 50, table on disk in
skani's row order), then the native greedy and dynamic selections on the rows.  N=8000: 32 M pairs."""
import json, os, shutil, sys, tempfile, time
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
import bench
from skder_amd import engine, synth, selection
from skder_amd.skder import Database
N = int(os.environ.get("N", "8000"))
ctx = engine.Context(0)
sk, paths, n50, t_sketch = bench.one_species_database(engine, ctx, torch, synth, N, 0)
t0 = time.perf_counter()
db = Database.from_sketches(sk, paths, n50, device=0)
sk.close()
t_db = time.perf_counter() - t0
tmp = tempfile.mkdtemp(prefix="skder_amd_one_species_greedy_")
try:
    out_tsv = os.path.join(tmp, "Skani_Triangle_Edge_Output.txt")
    t0 = time.perf_counter()
    rows = db.triangle(50.0, 89.5, out_tsv=out_tsv)
    t_tri = time.perf_counter() - t0
    table_bytes = os.path.getsize(out_tsv)
    os.remove(out_tsv)
    t0 = time.perf_counter()
    g = selection.native_greedy(rows, paths, n50, 99.5, 50.0)
    t_g = time.perf_counter() - t0
    t0 = time.perf_counter()
    d = selection.native_dynamic(rows, paths, n50, 99.5, 50.0, 10.0)
    t_d = time.perf_counter() - t0
    print(json.dumps({"workload": "%d genomes of ONE species (34 real C. granulosum assemblies + device-generated descendants): skani triangle --min-af 50 -s 89.5 "
                                  "replaced by skder_amd_db_triangle with the table on disk, then the native selections at -i 99.5 -f 50" % N,
                      "genomes": N, "pairs": N * (N - 1) // 2, "rows": int(len(rows)), "table_bytes": table_bytes,
                      "seconds": {"generate_and_sketch": t_sketch, "database_incl_index": t_db, "triangle_with_table_on_disk": t_tri,
                                  "greedy_selection": t_g, "dynamic_selection": t_d},
                      "us_per_pair_triangle_incl_table": 1e6 * t_tri / (N * (N - 1) / 2), "representatives_greedy": len(g), "representatives_dynamic": len(d),
                      "command": "N=%d python profiles/run/r4_one_species_greedy.py" % N}))
finally:
    shutil.rmtree(tmp, ignore_errors=True)
    db.close()
