#!/bin/bash
# round 4: the default bench line as the driver runs it, timed
mkdir -p gpurun_out/r4g
T0=$(date +%s.%N)
python bench.py > gpurun_out/r4g/bench.json 2> gpurun_out/r4g/bench.err
echo "rc $? wall $(echo "$(date +%s.%N) - $T0" | bc) s"
grep -iE "error|Traceback" gpurun_out/r4g/bench.err | head -5
python - <<PY
import json
d=json.loads(open("gpurun_out/r4g/bench.json").read().strip().splitlines()[-1])
print(d["value"]/1e6, d["ms_per_step"], d["roofline"]["frac"], d["cpu_baseline"]["value"], d["cpu_baseline"].get("measured_on_sample"))
r=d["realistic"]
for k,v in r.items():
    if isinstance(v,dict): print(k, {a:b for a,b in v.items() if a in ("us_per_chained_pair","slow_path_fraction","triangle_ms","pairs_per_s","seconds_speculative_batches","seconds_one_search_per_representative","listings_identical","rows_per_search","error","total_s_of_this_leg","chain_stage_ms")})
print(d.get("end_to_end",{}).get("extrapolated_full_workload_s"), d.get("end_to_end_gz",{}).get("extrapolated_full_workload_s"))
PY
