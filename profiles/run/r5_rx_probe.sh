#!/bin/bash
# timing only: run_extract_kernel with one more 16-byte stream per four seeds (13 B per seed instead of 9): does its time follow its bytes?
mkdir -p gpurun_out/r5rx
B="--steps 6 --warmup 2 --no-cpu-baseline --no-realistic --e2e-genomes 0 --parity-pairs 0 --low-mem-genomes 0 --one-species-genomes 0"
for v in 0 1 0 1; do
  cp skder_amd/lib_rx$v.so.bin skder_amd/libskder_amd.so
  SKDER_AMD_QUEUES=1 TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d gpurun_out/r5rx/kt_$v -o kt --output-format csv -- python bench.py $B > /dev/null 2>&1
  python - <<PY
import csv, glob
for f in glob.glob('gpurun_out/r5rx/kt_$v/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'run_extract' in r['Name']: print('probe $v', r['Calls'], r['AverageNs'])
PY
done
cp skder_amd/lib_rx0.so.bin skder_amd/libskder_amd.so
