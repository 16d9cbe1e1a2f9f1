#!/bin/bash
# round 4: the row kernel with the generalised stretch rule -- parity (normal, every chunk forced onto it), timing, counts
mkdir -p gpurun_out/r4d
K="not config4 and not config5 and not 1000_genomes and not properties_at_scale and not ranks_share and not rccl"
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$K" > gpurun_out/r4d/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r4d/pytest.log
SKDER_AMD_FORCE_SLOW=1 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "index_and_triangle or synthetic_with_screen or repeats_indels or repeat_rich or real_derived or benchmark_size or mixed_genome or degenerate or beyond_16 or repetitive_cutoff or dropin" > gpurun_out/r4d/pytest_force_slow.log 2>&1; echo "rc=$?" >> gpurun_out/r4d/pytest_force_slow.log
tail -n 3 gpurun_out/r4d/pytest.log gpurun_out/r4d/pytest_force_slow.log
D=${D:-8} python profiles/run/r3_real_debug.py 2>&1 | grep -E "kernels|us_per" | tail -2
cd skder_amd/csrc && touch chain_rows.hip && make EXTRA=-DSKDER_ROWS_STATS 2>&1 | grep -E "error" ; cd ../..
D=${D:-8} python profiles/run/r3_real_debug.py 2>&1 | grep -E "general path" | tail -2
