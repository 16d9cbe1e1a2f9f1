#!/bin/bash
# round 5: refill threshold once more (fat list items), then SQ counters of the chaining kernels on the real-structure workload
# (34 assemblies x D descendants), separate --pmc passes with --kernel-trace only
export TMPDIR=/tmp D=${D:-8}
for R in 32 24 16 12; do
  echo "== refill_min $R"
  SKDER_AMD_RUNS_REFILL=$R python profiles/run/r3_real_debug.py 2>&1 | grep -E "^\{" | python -c "
import sys,ast
r=ast.literal_eval(sys.stdin.read().strip().splitlines()[-1])
print({k:(round(v,3) if isinstance(v,float) else v) for k,v in r.items() if k in ('triangle_ms','chain_fast_ms','chain_slow_ms','us_per_chained_pair')})"
done
OUT=gpurun_out/r5pmc
mkdir -p $OUT
i=0
for G in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR"; do
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
        for name in ("chain_runs_kernel",'chain_rows_kernel','slow_wave_kernel','chain_single_kernel','finalize_kernel','run_extract_kernel','join_probe_kernel'):
            if name in k: tot[name][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in tot.items():
    lanes = v.get("SQ_THREAD_CYCLES_VALU",0)/max(v.get("SQ_ACTIVE_INST_VALU",1),1)
    print(k, {c: '%.4g' % x for c,x in sorted(v.items())}, 'lanes per VALU instruction %.1f of 64' % lanes)
PY
