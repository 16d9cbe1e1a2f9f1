#!/bin/bash
# round 5: the run loop with lanes that draw new chunks as they finish (chain_runs.hip): parity first, then the refill threshold
# on the real-structure set (34 assemblies x D descendants), each kernel alone on one queue
mkdir -p gpurun_out/r5c
K="index_and_triangle or synthetic_with_screen or repeats_indels or structural or repeat_rich or real_derived or benchmark_size or mixed_genome or degenerate or beyond_16 or repetitive_cutoff or dropin or small_batches or overflowed or anchor_in_reach"
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$K" > gpurun_out/r5c/pytest.log 2>&1; echo "parity rc=$?"; tail -n 2 gpurun_out/r5c/pytest.log
SKDER_AMD_NO_SIEVE=1 timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$K" > gpurun_out/r5c/pytest_no_sieve.log 2>&1; echo "no-sieve parity rc=$?"; tail -n 2 gpurun_out/r5c/pytest_no_sieve.log
for R in 64 48 32 24 16 8 4; do
  echo "== refill_min $R"
  SKDER_AMD_RUNS_REFILL=$R D=${D:-8} python profiles/run/r3_real_debug.py 2>&1 | grep -E "^\{" | python -c "
import sys,ast
r=ast.literal_eval(sys.stdin.read().strip().splitlines()[-1])
print({k:(round(v,3) if isinstance(v,float) else v) for k,v in r.items() if k in ('triangle_ms','chain_fast_ms','chain_slow_ms','join_ms','run_extract_ms','finalize_ms','us_per_chained_pair','slow_path_fraction','chained_pairs')})"
done
