#!/bin/bash
# round 3: the ladder form of the slow chaining path -- parity (normal and with EVERY chunk forced down the slow path), A/B timing
mkdir -p gpurun_out/r3i
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not config4 and not config5 and not 1000_genomes and not properties_at_scale and not ranks_share and not rccl" > gpurun_out/r3i/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r3i/pytest.log
SKDER_AMD_FORCE_SLOW=1 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "index_and_triangle or synthetic_with_screen or repeats_indels or structural or repeat_rich or real_derived or benchmark_size or mixed_genome or degenerate or beyond_16 or repetitive_cutoff or dropin" > gpurun_out/r3i/pytest_force_slow.log 2>&1; echo "rc=$?" >> gpurun_out/r3i/pytest_force_slow.log
tail -2 gpurun_out/r3i/pytest.log gpurun_out/r3i/pytest_force_slow.log
for v in plain ladders; do
  if [ $v = plain ]; then export SKDER_AMD_SLOW_PLAIN=1; else unset SKDER_AMD_SLOW_PLAIN; fi
  D=8 python profiles/run/r3_real_debug.py 2>&1 | grep -E "kernels" | tail -1
done
