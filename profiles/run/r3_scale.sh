#!/bin/bash
# round 3: scale checks on ONE MI355X (BASELINE configs 4 / 5 are 8-GPU configurations): 20,000 x 3 Mb, 20,000 x 1-8 Mb, 30,000 x 1-8 Mb
mkdir -p gpurun_out/r3s
timeout 600 python bench.py --genomes 20000 --steps 2 --warmup 2 --no-cpu-baseline --no-realistic --low-mem-genomes 0 --e2e-genomes 0 --batch-genomes 1250 > gpurun_out/r3s/scale_20000.json 2> gpurun_out/r3s/scale_20000.err; tail -c 300 gpurun_out/r3s/scale_20000.err
timeout 600 python bench.py --genomes 20000 --len-range 1000000 8000000 --steps 2 --warmup 2 --no-cpu-baseline --no-realistic --low-mem-genomes 0 --e2e-genomes 0 --batch-genomes 1250 > gpurun_out/r3s/scale_20000_mixed.json 2> gpurun_out/r3s/scale_20000_mixed.err; tail -c 300 gpurun_out/r3s/scale_20000_mixed.err
timeout 900 python bench.py --genomes 30000 --len-range 1000000 8000000 --steps 1 --warmup 2 --no-cpu-baseline --no-realistic --low-mem-genomes 0 --e2e-genomes 0 --batch-genomes 1250 > gpurun_out/r3s/scale_30000_mixed.json 2> gpurun_out/r3s/scale_30000_mixed.err; tail -c 300 gpurun_out/r3s/scale_30000_mixed.err
for f in scale_20000 scale_20000_mixed scale_30000_mixed; do python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r3s/$f.json").read().strip().splitlines()[-1]); print("$f", round(d["value"]/1e6,1), "M pairs/s", round(d["ms_per_step"],1), "ms", d["config"].get("chained_pairs"))
except Exception as e: print("$f failed", e)
PY
done
