#!/bin/bash
# profiles/collect.sh ROUND -- run on the MI355X box from the repository root (through gpurun):
#   kernel-trace statistics, FETCH_SIZE and WRITE_SIZE in separate --pmc passes (never combined with
#   other trace domains), the counter calibration of profiles/calib, and the default bench line.
# Raw output goes to gpurun_out/prof_$ROUND/; profiles/summarise.py condenses it into profiles/.
set -u
R=${1:-round1}
OUT=gpurun_out/prof_$R
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 bench.py --no-cpu-baseline --steps 1 --warmup 0"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats --output-format csv -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 > $OUT/stats.log 2>&1
# the chaining batches of a step alternate between two queues and overlap: a second trace with everything on one queue
# gives the chain-stage kernels' own durations (the counter passes below run that way as well)
export SKDER_AMD_QUEUES=1
rocprofv3 --kernel-trace --stats -d $OUT/stats1q -o stats --output-format csv -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 > $OUT/stats1q.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o fetch --output-format csv -- $BENCH > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o write --output-format csv -- $BENCH > $OUT/write.log 2>&1
if [ ! -x profiles/calib/fetch_calib ]; then hipcc --offload-arch=gfx950 -O3 -o profiles/calib/fetch_calib profiles/calib/fetch_calib.hip; fi
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/calib_fetch -o calib --output-format csv -- profiles/calib/fetch_calib > $OUT/calib_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/calib_write -o calib --output-format csv -- profiles/calib/fetch_calib > $OUT/calib_write.log 2>&1
unset SKDER_AMD_QUEUES
python3 bench.py > $OUT/bench_line.json 2> $OUT/bench.log
export SKDER_AMD_QUEUES=1
bash profiles/tools/pmc.sh join_probe_kernel > $OUT/pmc_join_probe_kernel.txt 2>&1
bash profiles/tools/pmc.sh run_extract_kernel > $OUT/pmc_run_extract_kernel.txt 2>&1
bash profiles/tools/pmc.sh chain_single_kernel > $OUT/pmc_chain_single_kernel.txt 2>&1
bash profiles/tools/pmc.sh sketch_tiles_kernel > $OUT/pmc_sketch_tiles_kernel.txt 2>&1
tail -c 600 $OUT/bench_line.json
