#!/bin/bash
# round 4: low_mem_greedy with the searches restricted to the database genomes that can still change the result -- parity tests, then the one-species workload
mkdir -p gpurun_out/r4m
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "speculative_search or config4 or driver_end_to_end or several_gpus or one_species" > gpurun_out/r4m/pytest.log 2>&1; tail -3 gpurun_out/r4m/pytest.log
N=5000 timeout 600 python profiles/run/r4_one_species.py > gpurun_out/r4m/one_species_5000.json 2> gpurun_out/r4m/err.log; tail -c 1200 gpurun_out/r4m/one_species_5000.json
