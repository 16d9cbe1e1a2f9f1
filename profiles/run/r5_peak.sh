#!/bin/bash
# round 5: the run loop settles paths whose best end is an earlier anchor when at most two anchors lie behind it: parity, decline causes, timing
mkdir -p gpurun_out/r5h
K="index_and_triangle or synthetic_with_screen or repeats_indels or structural or repeat_rich or real_derived or benchmark_size or mixed_genome or degenerate or beyond_16 or repetitive_cutoff or dropin or small_batches or overflowed or anchor_in_reach"
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$K" > gpurun_out/r5h/pytest.log 2>&1; echo "parity rc=$?"; tail -n 3 gpurun_out/r5h/pytest.log
SKDER_AMD_NO_SIEVE=1 timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$K" > gpurun_out/r5h/pytest_no_sieve.log 2>&1; echo "no-sieve parity rc=$?"; tail -n 3 gpurun_out/r5h/pytest_no_sieve.log
D=8 python profiles/run/r3_real_debug.py 2>&1 | grep -E "batch:|^\{" | tail -3 | cut -c1-600
