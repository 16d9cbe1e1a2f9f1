#!/bin/bash
# HISTORICAL (kept as the record of how a committed figure was measured): this script sets SKDER_AMD_SKETCH_VARIANT, a switch the library
# stopped reading in round 5 (include/skder_amd.h lists the live ones) -- on today's tree it would measure the default build under a
# variant's label.  To repeat the measurement check out the round it belongs to (r3_* : round 3, r4_* : round 4).
# round 3: the sketch body's instruction-selection table, parity of the chosen body, its effect on the step
mkdir -p gpurun_out/r3c
./profiles/calib/sketch_body_bench > gpurun_out/r3c/sketch_body.json
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sketch or index_and_triangle or partly or dropin or low_complexity or synthetic_with_screen or truth or degenerate or beyond_16" > gpurun_out/r3c/pytest_subset.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3c/pytest_subset.log
for v in 0 default; do
  if [ $v = default ]; then unset SKDER_AMD_SKETCH_VARIANT; else export SKDER_AMD_SKETCH_VARIANT=$v; fi
  python bench.py --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/r3c/bench_v$v.json 2> gpurun_out/r3c/bench_v$v.err
done
tail -3 gpurun_out/r3c/pytest_subset.log
