"""decline causes of the fast chaining path on the real-structure workload (34 assemblies x D descendants)"""
import os, sys, time
sys.path.insert(0, os.getcwd())
os.environ["SKDER_AMD_DEBUG"] = "1"
os.environ["SKDER_AMD_QUEUES"] = "1"
import numpy as np, torch
import bench
from skder_amd import engine
D = int(os.environ.get("D", "8"))
ctx = engine.Context(0)
gold = os.path.join("tests", "golden", "genomes")
recs = [bench._read_fasta_records(os.path.join(gold, n)) for n in sorted(os.listdir(gold))]
fam = [bench._real_descendant(1000 * a + d, recs[a][1], recs[a][0]) for a in range(len(recs)) for d in range(D)]
r = bench._triangle_stats(engine, ctx, torch, [g[1] for g in fam], [g[0] for g in fam], 80.0, steps=1)
print(r)
