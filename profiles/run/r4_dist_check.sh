#!/bin/bash
mkdir -p gpurun_out/r4h
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ranks_share or several_gpus or rccl" > gpurun_out/r4h/pytest_dist.log 2>&1; tail -5 gpurun_out/r4h/pytest_dist.log
bash profiles/run/r4_dist_overhead.sh 2>&1 | tee gpurun_out/r4h/dist2.log
