#!/bin/bash
# round 3: FASTA parsing on the device -- parity with the host reader, randomized drop-in run, ingest rates
mkdir -p gpurun_out/r3n
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "device_fasta or gzip or n50 or degenerate or listing_edge or dropin or store or config4 or several_gpus or driver_end" > gpurun_out/r3n/pytest.log 2>&1; tail -n 5 gpurun_out/r3n/pytest.log
timeout 200 python tests/tools/fuzz_dropin.py 8000000 8000400 > gpurun_out/r3n/fuzz_dropin.log 2>&1; echo "dropin: $(grep -c ' ok' gpurun_out/r3n/fuzz_dropin.log) ok, $(grep -c MISMATCH gpurun_out/r3n/fuzz_dropin.log) mismatches"; tail -n 2 gpurun_out/r3n/fuzz_dropin.log
