"""round 4: the reference's published workload in its real shape -- lowMemGreedyDerep -i 99.5 -f 50 on N genomes of ONE species
(real assemblies + device-generated descendants), from resident sketches; N=20000 by default (README.md:27: > 20,000 genomes, 2.25 h)"""
import json, os, sys
sys.path.insert(0, os.getcwd())
import torch
import bench
from skder_amd import engine, synth
N = int(os.environ.get("N", "20000"))
ctx = engine.Context(0)
r = bench.low_mem_greedy_one_species(engine, ctx, torch, synth, N, 0)
print(json.dumps(r))
