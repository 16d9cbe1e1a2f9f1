#!/bin/bash
# kernel trace of low_mem_greedy on N genomes of one species: which kernels, and how much of the wall time is the GPU busy?
N=${N:-5000}
mkdir -p gpurun_out/r5os
N=$N TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d gpurun_out/r5os/kt -o kt --output-format csv -- python profiles/run/r5_one_species.py > gpurun_out/r5os/out.json 2>gpurun_out/r5os/err.log
tail -c 400 gpurun_out/r5os/out.json
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r5os/kt/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('kernel time total %.2f s' % (tot / 1e9))
for r in rows[:14]: print('%-40s %6s %9.1f us %7.1f ms %5s %%' % (r['Name'][:40], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6, r['Percentage']))
t = glob.glob('gpurun_out/r5os/kt/**/*kernel_trace.csv', recursive=True)[0]
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in csv.DictReader(open(t)))
busy = 0; cs, ce = ev[0]
for s, e in ev[1:]:
    if s <= ce: ce = max(ce, e)
    else: busy += ce - cs; cs, ce = s, e
busy += ce - cs
print('span %.2f s, GPU busy (union) %.2f s' % ((ev[-1][1] - ev[0][0]) / 1e9, busy / 1e9))
PY
