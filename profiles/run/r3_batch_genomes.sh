#!/bin/bash
# round 3: genomes per resident input batch (one sketch call each) against the step
for b in 2500 5000 1250; do
  python bench.py --no-realistic --low-mem-genomes 0 --no-cpu-baseline --e2e-genomes 0 --batch-genomes $b 2>/dev/null > /tmp/bg_$b.json
  python - $b <<'PY'
import json, sys
d = json.loads(open("/tmp/bg_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d["value"] / 1e6, 1), round(d["ms_per_step"], 2), d["roofline"]["other_ms"], d["roofline"]["host_wall_ms"])
PY
done
