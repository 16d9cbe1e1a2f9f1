#!/bin/bash
# round 5: refill threshold of the run loop (work drawn in per-wavefront ranges), real-structure set, each kernel alone on one queue
mkdir -p gpurun_out/r5c
for R in 64 48 32 24 16 12 8; do
  echo "== refill_min $R"
  SKDER_AMD_RUNS_REFILL=$R D=${D:-8} python profiles/run/r3_real_debug.py 2>&1 | grep -E "^\{" | python -c "
import sys,ast
r=ast.literal_eval(sys.stdin.read().strip().splitlines()[-1])
print({k:(round(v,3) if isinstance(v,float) else v) for k,v in r.items() if k in ('triangle_ms','chain_fast_ms','chain_slow_ms','join_ms','run_extract_ms','finalize_ms','us_per_chained_pair','slow_path_fraction','chained_pairs')})"
done
cd skder_amd/csrc && touch chain_runs.hip chain.hip && make EXTRA=-DSKDER_RUNS_STATS 2>&1 | grep -E "error" ; cd ../..
for R in 64 16; do
SKDER_AMD_RUNS_REFILL=$R D=${D:-8} python profiles/run/r3_real_debug.py 2>&1 | grep -E "run loop:" | tail -1
done
