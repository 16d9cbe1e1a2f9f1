#!/bin/bash
# round 4: the row-per-chunk chaining kernel -- parity (normal mix and with EVERY chunk forced onto it), A/B timing on the real-structure set
mkdir -p gpurun_out/r4a
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not config4 and not config5 and not 1000_genomes and not properties_at_scale and not ranks_share and not rccl" > gpurun_out/r4a/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r4a/pytest.log
SKDER_AMD_FORCE_SLOW=1 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "index_and_triangle or synthetic_with_screen or repeats_indels or structural or repeat_rich or real_derived or benchmark_size or mixed_genome or degenerate or beyond_16 or repetitive_cutoff or dropin" > gpurun_out/r4a/pytest_force_slow.log 2>&1; echo "rc=$?" >> gpurun_out/r4a/pytest_force_slow.log
tail -3 gpurun_out/r4a/pytest.log gpurun_out/r4a/pytest_force_slow.log
for v in wave rows rows_all; do
  unset SKDER_AMD_NO_ROWS SKDER_AMD_FORCE_SLOW
  if [ $v = wave ]; then export SKDER_AMD_NO_ROWS=1; fi
  if [ $v = rows_all ]; then export SKDER_AMD_FORCE_SLOW=1; fi
  echo "== $v"
  D=${D:-8} python profiles/run/r3_real_debug.py 2>&1 | grep -E "kernels|triangle_ms" | tail -2
done
