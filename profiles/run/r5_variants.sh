#!/bin/bash
# variants of the library on one box, interleaved twice: kernel averages of the headline (rocprofv3 of the same bench command) and the
# real-structure us per pair.  usage: r5_variants.sh PREFIX KERNEL-SUBSTRING...   (libraries skder_amd/lib_PREFIX*.so.bin, built beforehand)
P=$1; shift; K="$*"
mkdir -p gpurun_out/r5var
B="--steps 10 --warmup 3 --no-cpu-baseline --no-realistic --e2e-genomes 0 --parity-pairs 0 --low-mem-genomes 0 --one-species-genomes 0"
cp skder_amd/libskder_amd.so /tmp/keep.so
for rep in 1 2; do for f in skder_amd/lib_$P*.so.bin; do
  v=$(basename $f .so.bin); cp $f skder_amd/libskder_amd.so
  r=$(D=30 python profiles/run/r3_real_debug.py 2>&1 | grep -E "^\{" | python -c "import ast,sys; d=ast.literal_eval(sys.stdin.read()); print(round(d.get('us_per_chained_pair'),4))")
  TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d gpurun_out/r5var/kt_${v}_$rep -o kt --output-format csv -- python bench.py $B > gpurun_out/r5var/b_${v}_$rep.json 2>/dev/null
  python - <<PY
import csv, glob, json
ms = json.load(open('gpurun_out/r5var/b_${v}_$rep.json'))['ms_per_step']
out = []
for f in glob.glob('gpurun_out/r5var/kt_${v}_$rep/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r['Name'] for k in "$K".split()): out.append('%s %.1f us' % (r['Name'][:22], float(r['AverageNs']) / 1e3))
print('$v', '; '.join(out), '; step ms (profiled)', round(ms, 2), '; real us/pair', $r)
PY
done; done
cp /tmp/keep.so skder_amd/libskder_amd.so
