#!/bin/bash
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
