#!/bin/bash
# finalize variants (threads per workgroup x bins) on one box, interleaved twice: kernel average on the headline, real-structure us per pair.
# The variants are built beforehand: for each "T B": make -C skder_amd/csrc EXTRA="-DFIN_THREADS=T -DFIN_BINS=B" (after touch chain.h);
#   cp skder_amd/libskder_amd.so skder_amd/lib_vT_B.so.bin
mkdir -p gpurun_out/r5fin
B="--steps 10 --warmup 3 --no-cpu-baseline --no-realistic --e2e-genomes 0 --parity-pairs 0 --low-mem-genomes 0 --one-species-genomes 0"
cp skder_amd/libskder_amd.so /tmp/keep.so
for rep in 1 2; do for f in skder_amd/lib_v*.so.bin; do
  v=$(basename $f .so.bin); cp $f skder_amd/libskder_amd.so
  r=$(D=30 python profiles/run/r3_real_debug.py 2>&1 | grep -E "^\{" | python -c "import ast,sys; d=ast.literal_eval(sys.stdin.read()); print(round(d.get('us_per_chained_pair'),4))")
  TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d gpurun_out/r5fin/kt_${v}_$rep -o kt --output-format csv -- python bench.py $B > gpurun_out/r5fin/b_${v}_$rep.json 2>/dev/null
  python - <<PY
import csv, glob, json
ms = json.load(open('gpurun_out/r5fin/b_${v}_$rep.json'))['ms_per_step']
for f in glob.glob('gpurun_out/r5fin/kt_${v}_$rep/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'finalize' in r['Name']: print('$v', 'finalize us', round(float(r['AverageNs'])/1e3,1), 'step ms (profiled)', round(ms,2), 'real us/pair', $r)
PY
done; done
cp /tmp/keep.so skder_amd/libskder_amd.so
