#!/bin/bash
# round 5: state of the tree at the start of the round -- GPU tests, then bench.py exactly as the driver runs it
mkdir -p gpurun_out/r5a
T0=$(date +%s)
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r5a/pytest.log 2>&1
echo "pytest rc $? after $(( $(date +%s) - T0 )) s"; tail -3 gpurun_out/r5a/pytest.log
T0=$(date +%s)
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r5a/bench.json 2> gpurun_out/r5a/bench.err
echo "bench rc $? after $(( $(date +%s) - T0 )) s"
python - <<PY
import json
d=json.loads(open("gpurun_out/r5a/bench.json").read().strip().splitlines()[-1])
print(d["value"]/1e6, d["ms_per_step"], d["roofline"].get("ms_per_step_one_queue"), d["roofline"].get("kernel_ms"))
PY
