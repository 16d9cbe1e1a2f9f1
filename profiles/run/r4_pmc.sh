#!/bin/bash
# HISTORICAL (kept as the record of how a committed figure was measured): this script sets SKDER_AMD_RRUNS, a switch the library
# stopped reading in round 5 (include/skder_amd.h lists the live ones) -- on today's tree it would measure the default build under a
# variant's label.  To repeat the measurement check out the round it belongs to (r3_* : round 3, r4_* : round 4).
# round 4: SQ counters of the chaining kernels on the real-structure workload (34 assemblies x D descendants), separate passes.
# TAG names the output; extra environment (SKDER_AMD_RRUNS=1, SKDER_AMD_NO_ROWS=1 ...) selects a variant
export TMPDIR=/tmp D=${D:-8}
TAG=${TAG:-r4}
OUT=gpurun_out/r4pmc/$TAG
mkdir -p $OUT
i=0
for G in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $G -d $OUT/p$i -o p --output-format csv -- python3 profiles/run/r3_real_prof.py > $OUT/p$i.log 2>&1
done
python3 - $OUT <<'PY' | tee $OUT/summary.txt
import csv,glob,sys,collections
out=sys.argv[1]
tot=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out+'/p*/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0].replace('void ','')
        for name in ("chain_runs_kernel",'chain_rruns_kernel','chain_rows_kernel','slow_wave_kernel','chain_single_kernel','finalize_kernel','run_extract_kernel','join_probe_kernel'):
            if name in k: tot[name][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in tot.items():
    print(k, {c: '%.4g' % x for c,x in sorted(v.items())})
PY
