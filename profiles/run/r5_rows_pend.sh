#!/bin/bash
# round 5: rows kernel with a chunk's chains buffered in LDS (one atomic per chunk), run loop with one pair_na atomic per pair and service: parity, timing
export TMPDIR=/tmp D=8
K="index_and_triangle or synthetic_with_screen or structural or repeat_rich or real_derived or dropin or small_batches or repeats_indels or anchor_in_reach or overflowed or benchmark_size or mixed_genome or degenerate or repetitive_cutoff"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$K" 2>&1 | tail -1
SKDER_AMD_FORCE_SLOW=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$K" 2>&1 | tail -1
for i in 1 2; do python profiles/run/r3_real_debug.py 2>&1 | grep -E "^\{" | tail -1 | python -c "
import sys,ast
r=ast.literal_eval(sys.stdin.read().strip())
print({k:round(v,3) for k,v in r.items() if k in ('chain_fast_ms','chain_slow_ms','us_per_chained_pair','triangle_ms')})"; done
TAG=pend bash profiles/run/r5_kt.sh | grep -E "chain_rows|chain_runs|chain_single"
