#!/bin/bash
# round 3: the join's two probes side by side (per-kernel HIP-event durations of bench.py, one queue), after the parity tests
mkdir -p gpurun_out/r3p
python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r3p/pytest.log 2>&1; tail -n 3 gpurun_out/r3p/pytest.log
for v in v2 v1; do
  if [ $v = v1 ]; then export SKDER_AMD_JOIN_V1=1; else unset SKDER_AMD_JOIN_V1; fi
  python bench.py --no-realistic --low-mem-genomes 0 --no-cpu-baseline --e2e-genomes 0 > gpurun_out/r3p/bench_$v.json 2> gpurun_out/r3p/bench_$v.err
  python -c "
import json; d=json.loads(open('gpurun_out/r3p/bench_$v.json').read().strip().splitlines()[-1]); print('$v', round(d['value']/1e6,1), 'M pairs/s', round(d['ms_per_step'],2), 'ms', d['roofline']['kernel_ms'])"
done
