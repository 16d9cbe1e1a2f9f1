#!/bin/bash
mkdir -p gpurun_out/r5b
cp tests/golden/full_size_digests.json gpurun_out/r5b/full_size_digests.json
SKDER_AMD_WRITE_DIGESTS=gpurun_out/r5b/full_size_digests.json timeout 1500 python -m pytest tests/test_gpu_full_size.py -m gpu -x -q -rs --durations=5 -k config5 > gpurun_out/r5b/pytest_cfg5.log 2>&1
echo "config5 rc $?"; tail -12 gpurun_out/r5b/pytest_cfg5.log
