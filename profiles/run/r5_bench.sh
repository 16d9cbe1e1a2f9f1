#!/bin/bash
# round 5: the default bench line as the driver runs it, timed
mkdir -p gpurun_out/r5g
T0=$(date +%s.%N)
python bench.py --steps 20 --warmup 5 > gpurun_out/r5g/bench.json 2> gpurun_out/r5g/bench.err
echo "rc $? wall $(echo "$(date +%s.%N) - $T0" | bc) s"
grep -iE "error|Traceback" gpurun_out/r5g/bench.err | head -5
python - <<PY
import json
d=json.loads(open("gpurun_out/r5g/bench.json").read().strip().splitlines()[-1])
print(d["value"]/1e6, d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["one_queue_steps_ms"], d["roofline"]["kernel_ms"], d["roofline"]["other_ms"])
print("config", {k:v for k,v in d["config"].items() if k!="workload"})
print("parity_sample", d.get("parity_sample",{}).get("mismatches"), "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("measured_on_sample",{}).get("pairs_per_s"))
r=d["realistic"]
for k,v in r.items():
    if isinstance(v,dict): print(k, {a:b for a,b in v.items() if a in ("us_per_chained_pair","slow_path_fraction","triangle_ms","pairs_per_s","seconds_speculative_batches","seconds_one_search_per_representative","listings_identical","rows_per_search","error","total_s_of_this_leg","chain_stage_ms","ms_per_step")})
PY
