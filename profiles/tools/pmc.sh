#!/bin/bash
# profiles/tools/pmc.sh KERNEL_SUBSTR  -- SQ counters for one kernel (separate passes, kernel-trace only)
export TMPDIR=/tmp
K=${1:-chain_fast_kernel}
OUT=gpurun_out/pmc_$K
mkdir -p $OUT
BENCH="python3 bench.py --no-cpu-baseline --steps 1 --warmup 0"
i=0
for G in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_I8 GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $G -d $OUT/p$i -o p --output-format csv -- $BENCH > $OUT/p$i.log 2>&1
done
python3 - "$K" $OUT <<'PY'
import csv,glob,sys,collections
k,out=sys.argv[1],sys.argv[2]
tot=collections.defaultdict(float); n=collections.defaultdict(int)
for f in glob.glob(out+'/p*/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if k in r['Kernel_Name']:
            tot[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
for c in sorted(tot): print(c, n[c], tot[c])
PY
