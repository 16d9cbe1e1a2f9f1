#!/bin/bash
# A/B on one box: two builds of the library, same bench command under rocprofv3, twice each, interleaved.
# The two builds are made beforehand (they travel with the snapshot, git-ignored):
#   git stash; make -C skder_amd/csrc; cp skder_amd/libskder_amd.so skder_amd/lib_old.so.bin; git stash pop
#   make -C skder_amd/csrc; cp skder_amd/libskder_amd.so skder_amd/lib_new.so.bin
mkdir -p gpurun_out/r5fin
B="--steps 10 --warmup 3 --no-cpu-baseline --no-realistic --e2e-genomes 0 --parity-pairs 0 --low-mem-genomes 0 --one-species-genomes 0"
for rep in 1 2; do for v in old new; do
  cp skder_amd/lib_$v.so.bin skder_amd/libskder_amd.so
  python bench.py $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['config'].get('real_derived_us_per_chained_pair'))"
  D=30 python profiles/run/r3_real_debug.py 2>&1 | grep -E "^\{" | python -c "import ast,sys; d=ast.literal_eval(sys.stdin.read()); print('  $v real', d.get('ms_per_triangle'), d.get('us_per_chained_pair'))"
  TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d gpurun_out/r5fin/kt_${v}_$rep -o kt --output-format csv -- python bench.py $B > /dev/null 2>&1
  python - <<PY
import csv, glob
for f in glob.glob('gpurun_out/r5fin/kt_${v}_$rep/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r['Name'] for k in ('finalize','chain_single','chain_runs')): print('  $v', r['Name'][:30], r['Calls'], r['AverageNs'], r['TotalDurationNs'])
PY
done; done
cp skder_amd/lib_new.so.bin skder_amd/libskder_amd.so
