"""round 5: the reference's published workload in its real shape -- lowMemGreedyDerep -i 99.5 -f 50 on N genomes of ONE species (real
assemblies + device-generated descendants), from resident sketches; N=20000 by default (README.md:27: > 20,000 genomes, 2.25 h).
SEQ=1 also runs the call-for-call loop (one search + TSV + parse per representative: ~9 minutes at N=20000)."""
import json, os, sys
sys.path.insert(0, os.getcwd())
import torch
import bench
from skder_amd import engine, synth
N = int(os.environ.get("N", "20000"))
ctx = engine.Context(0)
r = bench.low_mem_greedy_one_species(engine, ctx, torch, synth, N, 0, sequential=os.environ.get("SEQ") == "1")
print(json.dumps(r))
