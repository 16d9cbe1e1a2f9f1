#!/bin/bash
# round 5: fat list items in the run loop + the component-owner exchange: parity, ranks sharing the GPU, timing on the real-structure set
mkdir -p gpurun_out/r5f
K="index_and_triangle or synthetic_with_screen or repeats_indels or structural or repeat_rich or real_derived or benchmark_size or mixed_genome or degenerate or beyond_16 or repetitive_cutoff or dropin or small_batches or overflowed or anchor_in_reach or ranks_share"
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$K" > gpurun_out/r5f/pytest.log 2>&1; echo "parity rc=$?"; tail -n 12 gpurun_out/r5f/pytest.log
SKDER_AMD_NO_SIEVE=1 timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$K and not ranks_share" > gpurun_out/r5f/pytest_no_sieve.log 2>&1; echo "no-sieve parity rc=$?"; tail -n 2 gpurun_out/r5f/pytest_no_sieve.log
D=8 python profiles/run/r3_real_debug.py 2>&1 | grep -E "^\{" | tail -1
TAG=fat bash profiles/run/r5_kt.sh | head -8
