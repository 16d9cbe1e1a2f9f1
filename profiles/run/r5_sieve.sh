#!/bin/bash
# round 5: the sieve with a workgroup's chunks dealt to its lanes by record count: parity, then the headline's and the real-structure set's times, A/B
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "index_and_triangle or synthetic_with_screen or structural or repeat_rich or real_derived or dropin or small_batches or repeats_indels or anchor_in_reach or overflowed or benchmark_size or mixed_genome or degenerate" 2>&1 | tail -1
for V in 1 0; do
  (cd skder_amd/csrc && touch chain_extract.hip && make EXTRA=-DSIEVE_SORT=$V 2>&1 | grep -E "error")
  echo "== SIEVE_SORT=$V"
  python bench.py --steps 10 --warmup 3 --no-realistic --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']/1e6,2), round(d['ms_per_step'],2), d['roofline']['kernel_ms'])"
  D=8 python profiles/run/r3_real_debug.py 2>&1 | grep -E "^\{" | tail -1 | python -c "
import sys,ast
r=ast.literal_eval(sys.stdin.read().strip())
print({k:round(v,3) for k,v in r.items() if k in ('chain_fast_ms','chain_slow_ms','us_per_chained_pair')})"
  TAG=sieve$V bash profiles/run/r5_kt.sh | grep -E "chain_single|chain_runs"
done
