#!/bin/bash
# round 3: join timing (bench.py's one-queue HIP-event durations) + its SQ counters, for the kernel as built
mkdir -p gpurun_out/r3p
export TMPDIR=/tmp
python bench.py --no-realistic --low-mem-genomes 0 --no-cpu-baseline --e2e-genomes 0 > gpurun_out/r3p/bench_q.json 2> gpurun_out/r3p/bench_q.err
python -c "
import json; d=json.loads(open('gpurun_out/r3p/bench_q.json').read().strip().splitlines()[-1]); print(round(d['value']/1e6,1), 'M pairs/s', round(d['ms_per_step'],2), 'ms', d['roofline']['kernel_ms'])"
if [ -n "$PMC" ]; then
SKDER_AMD_QUEUES=1 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE -d gpurun_out/r3p/pq -o p --output-format csv -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 > gpurun_out/r3p/pq.log 2>&1
python3 - <<'PY'
import csv,glob,collections
tot=collections.defaultdict(float)
for f in glob.glob('gpurun_out/r3p/pq/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'join_probe' in r['Kernel_Name']: tot[r['Counter_Name']]+=float(r['Counter_Value'])
print(dict(tot))
PY
fi
