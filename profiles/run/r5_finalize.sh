#!/bin/bash
# round 5: finalize with the chunk marks in LDS (the bin counters' words), the chunk seed counts requested at the top and one barrier per
# round of the overlap filter.  Default build: GPU suite + bench line; then a build that sends every pair through the global marks: parity tests.
mkdir -p gpurun_out/r5fin
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py > gpurun_out/r5fin/bench.json 2> gpurun_out/r5fin/bench.err; tail -c 600 gpurun_out/r5fin/bench.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r5fin/bench.json'))
print(d['value'], d['ms_per_step'], d['config'].get('parity_sample_mismatches'), d['config'].get('real_derived_us_per_chained_pair'))
PY
TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d gpurun_out/r5fin/kt -o kt -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-realistic --e2e-genomes 0 --parity-pairs 0 --low-mem-genomes 0 --one-species-genomes 0 > /dev/null 2>&1
python - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/r5fin/kt/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r['Name'] for k in ('finalize', 'chain_', 'run_extract', 'join')): print(r['Name'][:40], r['Calls'], r['AverageNs'], r['TotalDurationNs'])
PY
(cd skder_amd/csrc && touch chain_finalize.hip && make EXTRA=-DFIN_LDS_MARKS=8 2>&1 | grep -E "error")
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
(cd skder_amd/csrc && touch chain_finalize.hip && make 2>&1 | grep -E "error")
