"""the real-structure workload of bench.py (34 real assemblies x D descendants, every pair chained) alone, for rocprofv3 --kernel-trace"""
import json, os, sys, time
sys.path.insert(0, os.getcwd())
from concurrent.futures import ThreadPoolExecutor
import bench, torch
from skder_amd import engine
ctx = engine.Context(0)
D = int(os.environ.get("D", "30"))
gold = os.path.join(bench.ROOT, "tests", "golden", "genomes")
recs = [bench._read_fasta_records(os.path.join(gold, n)) for n in sorted(os.listdir(gold))]
jobs = [(1000 * a + d, recs[a][1], recs[a][0]) for a in range(len(recs)) for d in range(D)]
with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
    fam = list(ex.map(lambda j: bench._real_descendant(*j), jobs))
r = bench._triangle_stats(engine, ctx, torch, [g[1] for g in fam], [g[0] for g in fam], 80.0, steps=int(os.environ.get("STEPS", "1")))
print(json.dumps(r))
