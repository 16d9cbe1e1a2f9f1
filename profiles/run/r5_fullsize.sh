#!/bin/bash
# round 5: the full-size tests (configs 3, 4, 5) with their digests recorded, then the bench line
mkdir -p gpurun_out/r5b
T0=$(date +%s)
SKDER_AMD_WRITE_DIGESTS=gpurun_out/r5b/full_size_digests.json timeout 1500 python -m pytest tests/test_gpu_full_size.py -m gpu -x -q --durations=5 > gpurun_out/r5b/pytest_full.log 2>&1
echo "full-size rc $? after $(( $(date +%s) - T0 )) s"; tail -12 gpurun_out/r5b/pytest_full.log
cp gpurun_out/r5b/full_size_digests.json tests/golden/full_size_digests.json
T0=$(date +%s)
timeout 1500 python -m pytest tests/test_gpu_full_size.py -m gpu -x -q > gpurun_out/r5b/pytest_full2.log 2>&1
echo "full-size against digests rc $? after $(( $(date +%s) - T0 )) s"; tail -3 gpurun_out/r5b/pytest_full2.log
T0=$(date +%s)
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r5b/bench.json 2> gpurun_out/r5b/bench.err
echo "bench rc $? after $(( $(date +%s) - T0 )) s"; tail -3 gpurun_out/r5b/bench.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r5b/bench.json").read().strip().splitlines()[-1])
print(d["value"]/1e6, d["ms_per_step"], d["roofline"].get("ms_per_step_one_queue"), d["roofline"].get("one_queue_steps_ms"), d["roofline"].get("kernel_ms"), d["roofline"]["host_wall_ms"])
print(d.get("parity_sample"))
PY
