#!/bin/bash
# HISTORICAL (kept as the record of how a committed figure was measured): this script sets SKDER_AMD_SKETCH_VARIANT, a switch the library
# stopped reading in round 5 (include/skder_amd.h lists the live ones) -- on today's tree it would measure the default build under a
# variant's label.  To repeat the measurement check out the round it belongs to (r3_* : round 3, r4_* : round 4).
# round 3, first GPU call: parity of the new sketch body + partly-indexed guard, A/B of the sketch variants
set -x
mkdir -p gpurun_out/r3a
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sketch or index_and_triangle or partly or dropin or low_complexity or synthetic_with_screen" > gpurun_out/r3a/pytest_subset.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3a/pytest_subset.log
for v in 0 1; do
  SKDER_AMD_SKETCH_VARIANT=$v python bench.py --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/r3a/bench_v$v.json 2> gpurun_out/r3a/bench_v$v.err
done
tail -3 gpurun_out/r3a/pytest_subset.log
python - <<'P'
import json
for v in (0,1):
    try:
        d=json.load(open('gpurun_out/r3a/bench_v%d.json'%v))
        print(v, d['ms_per_step'], d['roofline']['kernel_ms'])
    except Exception as e: print(v,'failed',e)
P
