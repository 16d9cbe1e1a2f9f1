#!/bin/bash
# Does index_genome_lds_kernel gain from TWO workgroups per CU?  Probe before rewriting its LDS layout: on 1.5 Mb genomes the present
# layout takes 56 KB, so two workgroups fit as soon as the registers allow it (-DIDXF_WAVES=8 / -DIDXF_U=4 builds: 64 VGPRs); the 3 Mb
# workload (96 KB: one workgroup per CU whatever the registers) is the control.  Kernel durations from rocprofv3 --kernel-trace --stats.
export TMPDIR=/tmp
OUT=gpurun_out/r6/idxocc; mkdir -p $OUT
B="--steps 4 --warmup 1 --no-cpu-baseline --no-realistic --e2e-genomes 0 --parity-pairs 0 --low-mem-genomes 0 --one-species-genomes 0"
for v in "$@"; do
  cp skder_amd/lib_$v.so.bin skder_amd/libskder_amd.so
  for w in "10000 1500000" "5000 3000000"; do
    set -- $w
    tag=${v}_$1
    rocprofv3 --kernel-trace --stats -d $OUT/$tag -o kt --output-format csv -- python3 bench.py $B --genomes $1 --genome-len $2 > $OUT/$tag.log 2>&1
    python3 - $tag $OUT/$tag <<'PY'
import csv, glob, sys
tag, d = sys.argv[1], sys.argv[2]
for f in glob.glob(d + '/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'index_genome' in r['Name'] or 'screen_rows' in r['Name']:
            print('%-22s %-32s calls %s avg %.3f ms' % (tag, r['Name'][:32], r['Calls'], float(r['AverageNs']) / 1e6))
PY
  done
done
