#!/bin/bash
# round 3: join_probe_kernel at two workgroups per CU (SGPR limit), pair index per chunk instead of a binary search
mkdir -p gpurun_out/r3k
python bench.py --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/r3k/bench.json 2>gpurun_out/r3k/bench.err
python -c "
import json; d=json.load(open('gpurun_out/r3k/bench.json')); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['ms_per_step_one_queue'])"
D=8 python profiles/run/r3_real_debug.py 2>&1 | grep -E "kernels" | tail -1
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "index_and_triangle or structural or repeat_rich or real_derived or small_batches or mixed_genome or benchmark_size" 2>&1 | tail -2
