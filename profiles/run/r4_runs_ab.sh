#!/bin/bash
# round 4: run loop A/B on the real-structure set (lists per record-count class): branching (v1) against predicated (v2), then round statistics
# (historical: the variant this script selected was measured and removed -- DESIGN.md section 8; it documents how the number was taken and no longer switches anything)
for v in v1 v2; do
  unset SKDER_AMD_RUNS_V1
  if [ $v = v1 ]; then export SKDER_AMD_RUNS_V1=1; fi
  echo "== run loop $v"
  D=${D:-8} python profiles/run/r3_real_debug.py 2>&1 | grep -E "kernels" | tail -1
done
unset SKDER_AMD_RUNS_V1
cd skder_amd/csrc && touch chain_runs.hip chain.hip && make EXTRA=-DSKDER_RUNS_STATS 2>&1 | grep -E "error" ; cd ../..
D=${D:-8} python profiles/run/r3_real_debug.py 2>&1 | grep -E "run loop:" | tail -2
