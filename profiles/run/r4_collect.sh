#!/bin/bash
# round 4: the judged profile set -- profiles/collect.sh round4 (kernel trace as shipped and on one queue, FETCH/WRITE passes + calibration,
# SQ counters of the headline's kernels, the default bench line), then the real-structure workload: kernel trace + SQ counters of the chaining
# kernels (the row kernel, the run loop), and the full GPU test suite
bash profiles/collect.sh round4 > gpurun_out/collect_round4.log 2>&1
tail -3 gpurun_out/collect_round4.log
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_round4/real
D=30 STEPS=1 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_round4/real/stats -o stats --output-format csv -- python3 profiles/run/r3_real_prof.py > gpurun_out/prof_round4/real/stats.log 2>&1
find gpurun_out/prof_round4/real/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/prof_round4/real_derived_kernel_stats.csv
D=8 TAG=round4 bash profiles/run/r4_pmc.sh > gpurun_out/prof_round4/real_pmc.txt 2>&1
tail -4 gpurun_out/prof_round4/real_pmc.txt | cut -c1-300
python -m pytest tests/ -x -q -m gpu > gpurun_out/pytest_round4.log 2>&1; tail -3 gpurun_out/pytest_round4.log
