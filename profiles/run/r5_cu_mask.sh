#!/bin/bash
# round 5: how do the chain stage's kernels scale with the compute units they get?  A measurement build confines the context's main queue to
# N CUs (hipExtStreamCreateWithCUMask); the headline step on ONE queue, per-kernel event times.  If the run extraction (HBM-bound) keeps its time
# on a fraction of the CUs, it could run in the join's shadow on CUs of its own.
(cd skder_amd/csrc && touch api.hip && make EXTRA=-DSKDER_CU_MASK_PROBE 2>&1 | grep -E "error")
for M in "" "256" "192" "128" "96" "64" "128 spread" "64 spread"; do
  set -- $M
  echo "== CUs ${1:-all (unmasked)} ${2:-}"
  SKDER_AMD_CU_MASK=$1 SKDER_AMD_CU_MASK_MODE=$2 SKDER_AMD_QUEUES=1 python bench.py --steps 3 --warmup 1 --no-realistic --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print(round(d['ms_per_step'],2), {k:round(v,2) for k,v in r['kernel_ms'].items()}, {k:round(v,2) for k,v in r['other_ms'].items()})"
done
(cd skder_amd/csrc && touch api.hip && make 2>&1 | grep -E "error")
