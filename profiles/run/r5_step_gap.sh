#!/bin/bash
# what does the host do while the GPU idles between two steps (the 1.2-1.5 ms gap behind a step's last kernel)?  HIP API trace + kernel trace
mkdir -p gpurun_out/r5gap
B="--steps 4 --warmup 2 --no-cpu-baseline --no-realistic --e2e-genomes 0 --parity-pairs 0 --low-mem-genomes 0 --one-species-genomes 0"
TMPDIR=/tmp rocprofv3 --hip-trace --kernel-trace -d gpurun_out/r5gap/t -o t --output-format csv -- python bench.py $B > gpurun_out/r5gap/bench.json 2>gpurun_out/r5gap/err.log
ls gpurun_out/r5gap/t
