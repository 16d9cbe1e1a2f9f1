#!/bin/bash
# round 5: parity of the run loop as it stands (per-wavefront draws, buffered decline list), decline causes, timing on the real-structure set
mkdir -p gpurun_out/r5e
K="index_and_triangle or synthetic_with_screen or repeats_indels or structural or repeat_rich or real_derived or benchmark_size or mixed_genome or degenerate or beyond_16 or repetitive_cutoff or dropin or small_batches or overflowed or anchor_in_reach or properties_at_scale"
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$K" > gpurun_out/r5e/pytest.log 2>&1; echo "parity rc=$?"; tail -n 2 gpurun_out/r5e/pytest.log
SKDER_AMD_NO_SIEVE=1 timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$K" > gpurun_out/r5e/pytest_no_sieve.log 2>&1; echo "no-sieve parity rc=$?"; tail -n 2 gpurun_out/r5e/pytest_no_sieve.log
D=8 python profiles/run/r3_real_debug.py 2>&1 | grep -E "batch:|^\{" | tail -3
