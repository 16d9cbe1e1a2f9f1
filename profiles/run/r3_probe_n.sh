#!/bin/bash
# A/B: bucket entries the join compares without a loop
mkdir -p gpurun_out/r3l
for n in 2 3 4; do
  cp exp/lib_probe$n.so skder_amd/libskder_amd.so
  python bench.py --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/r3l/bench_p$n.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/r3l/bench_p$n.json')); print($n, d['ms_per_step'], d['roofline']['kernel_ms']['join_probe_kernel'], d['roofline']['kernel_ms_two_queues']['join_probe_kernel'])"
done
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "index_and_triangle or structural or repeat_rich or mixed_genome or benchmark_size or repetitive" 2>&1 | tail -2
