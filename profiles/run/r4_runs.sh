#!/bin/bash
# round 4: the predicated run loop -- parity (normal mix, and every chunk with hits through the run loop), A/B timing on the real-structure set
# (historical: the variant this script selected was measured and removed -- DESIGN.md section 8; it documents how the number was taken and no longer switches anything)
mkdir -p gpurun_out/r4b
K="not config4 and not config5 and not 1000_genomes and not properties_at_scale and not ranks_share and not rccl"
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$K" > gpurun_out/r4b/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r4b/pytest.log
SKDER_AMD_NO_SIEVE=1 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "index_and_triangle or synthetic_with_screen or repeats_indels or structural or repeat_rich or real_derived or benchmark_size or mixed_genome or degenerate or beyond_16 or repetitive_cutoff or dropin" > gpurun_out/r4b/pytest_no_sieve.log 2>&1; echo "rc=$?" >> gpurun_out/r4b/pytest_no_sieve.log
tail -n 3 gpurun_out/r4b/pytest.log gpurun_out/r4b/pytest_no_sieve.log
for v in v1 v2; do
  unset SKDER_AMD_RUNS_V1
  if [ $v = v1 ]; then export SKDER_AMD_RUNS_V1=1; fi
  echo "== run loop $v"
  D=${D:-8} python profiles/run/r3_real_debug.py 2>&1 | grep -E "kernels|batch:" | tail -2
done
