#!/bin/bash
# round 5: the ring look-back of the run loop in straight-line form: parity, time on the real-structure set, lanes per VALU instruction
export TMPDIR=/tmp D=8
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "index_and_triangle or structural or repeat_rich or real_derived or dropin or small_batches or repeats_indels or anchor_in_reach or overflowed" 2>&1 | tail -1
SKDER_AMD_NO_SIEVE=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "index_and_triangle or structural or repeat_rich or real_derived or repeats_indels" 2>&1 | tail -1
for i in 1 2; do python profiles/run/r3_real_debug.py 2>&1 | grep -E "^\{" | tail -1 | python -c "
import sys,ast
r=ast.literal_eval(sys.stdin.read().strip())
print({k:round(v,3) for k,v in r.items() if k in ('chain_fast_ms','chain_slow_ms','us_per_chained_pair')})"; done
OUT=gpurun_out/r5try; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $OUT/p3 -o p --output-format csv -- python3 profiles/run/r3_real_prof.py > $OUT/p3.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD -d $OUT/p1 -o p --output-format csv -- python3 profiles/run/r3_real_prof.py > $OUT/p1.log 2>&1
python3 - $OUT <<'PY'
import csv,glob,sys,collections
tot=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1]+'/p*/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0].replace('void ','')
        if 'chain_runs_kernel' in k: tot['chain_runs_kernel'][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in tot.items():
    print(k, {c:'%.4g'%x for c,x in sorted(v.items())}, 'lanes %.1f' % (v['SQ_THREAD_CYCLES_VALU']/v['SQ_ACTIVE_INST_VALU']))
PY
