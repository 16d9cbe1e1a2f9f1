#!/bin/bash
# round 4: the second sieve -- parity (normal mix; every chunk with hits past the first sieve), A/B timing on the real-structure set
# (historical: the variant this script selected was measured and removed -- DESIGN.md section 8; it documents how the number was taken and no longer switches anything)
mkdir -p gpurun_out/r4c
K="not config4 and not config5 and not 1000_genomes and not properties_at_scale and not ranks_share and not rccl"
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$K" > gpurun_out/r4c/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r4c/pytest.log
SKDER_AMD_NO_SIEVE=1 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "index_and_triangle or synthetic_with_screen or repeats_indels or structural or repeat_rich or real_derived or benchmark_size or mixed_genome or degenerate or beyond_16 or repetitive_cutoff or dropin" > gpurun_out/r4c/pytest_no_sieve.log 2>&1; echo "rc=$?" >> gpurun_out/r4c/pytest_no_sieve.log
tail -n 3 gpurun_out/r4c/pytest.log gpurun_out/r4c/pytest_no_sieve.log
for v in nopaths paths; do
  unset SKDER_AMD_NO_PATHS
  if [ $v = nopaths ]; then export SKDER_AMD_NO_PATHS=1; fi
  echo "== $v"
  D=${D:-8} python profiles/run/r3_real_debug.py 2>&1 | grep -E "kernels|batch:|us_per" | tail -3
done
