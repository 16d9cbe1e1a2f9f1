#!/bin/bash
mkdir -p gpurun_out/r4h
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "clustered_truth or driver_end_to_end" > gpurun_out/r4h/pytest.log 2>&1; tail -5 gpurun_out/r4h/pytest.log
bash profiles/run/r4_dist_overhead.sh 2>&1 | tee gpurun_out/r4h/dist.log
