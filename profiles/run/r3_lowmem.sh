#!/bin/bash
# round 3: the reference's published workload shape (README.md:27) -- low_mem_greedy on 20,000 genomes, one MI355X
mkdir -p gpurun_out/r3d
python - > gpurun_out/r3d/low_mem_$N.json 2> gpurun_out/r3d/low_mem_$N.err <<'P'
import json, sys, os
sys.path.insert(0, os.getcwd())
import bench, torch
from skder_amd import engine, synth
ctx = engine.Context(0)
print(json.dumps(bench.low_mem_greedy_scale(engine, ctx, torch, synth, int(os.environ.get("N", "2000")), 2_800_000, 0)))
P
tail -c 1500 gpurun_out/r3d/low_mem_$N.json; tail -5 gpurun_out/r3d/low_mem_$N.err
