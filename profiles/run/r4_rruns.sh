#!/bin/bash
# round 4: the run DP in row form (chain_rruns.hip) -- parity on the normal mix and with every sieve-left chunk through it (SKDER_AMD_RRUNS=2), then A/B timing
# on the real-structure set: 0 = run loop -> general kernel, 1 = run loop -> row form -> general kernel, 2 = row form -> general kernel
mkdir -p gpurun_out/r4i
K="not config4 and not config5 and not 1000_genomes and not properties_at_scale and not ranks_share and not rccl and not several_gpus and not one_species"
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$K" > gpurun_out/r4i/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r4i/pytest.log
SKDER_AMD_RRUNS=2 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$K" > gpurun_out/r4i/pytest_rr2.log 2>&1; echo "rc=$?" >> gpurun_out/r4i/pytest_rr2.log
SKDER_AMD_RRUNS=2 SKDER_AMD_NO_SIEVE=1 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "index_and_triangle or synthetic_with_screen or repeats_indels or structural or repeat_rich or real_derived or benchmark_size or mixed_genome or degenerate or beyond_16 or repetitive_cutoff or dropin" > gpurun_out/r4i/pytest_rr2_nosieve.log 2>&1; echo "rc=$?" >> gpurun_out/r4i/pytest_rr2_nosieve.log
tail -n 4 gpurun_out/r4i/pytest.log gpurun_out/r4i/pytest_rr2.log gpurun_out/r4i/pytest_rr2_nosieve.log
for v in 0 1 2; do
  echo "== SKDER_AMD_RRUNS=$v"
  SKDER_AMD_RRUNS=$v D=${D:-8} python profiles/run/r3_real_debug.py 2>&1 | grep -E "kernels|run DP|batch:" | tail -5
done
