#!/bin/bash
# HISTORICAL (kept as the record of how a committed figure was measured): this script sets SKDER_AMD_ROWS_STOP, a switch the library
# stopped reading in round 5 (include/skder_amd.h lists the live ones) -- on today's tree it would measure the default build under a
# variant's label.  To repeat the measurement check out the round it belongs to (r3_* : round 3, r4_* : round 4).
# round 4: where the row kernel's time goes -- counts (stats build of chain_rows.hip) and the kernel stopped after each phase (timing only)
cd skder_amd/csrc && touch chain_rows.hip && make EXTRA=-DSKDER_ROWS_STATS 2>&1 | grep -E "error" ; cd ../..
D=${D:-8} python profiles/run/r3_real_debug.py 2>&1 | grep -E "general path|kernels" | tail -3
cd skder_amd/csrc && touch chain_rows.hip && make 2>&1 | grep -E "error" ; cd ../..
for stop in 1 2 3 0; do
  echo "== stop after phase $stop (0: whole kernel)"
  SKDER_AMD_ROWS_STOP=$stop D=${D:-8} python profiles/run/r3_real_debug.py 2>&1 | grep -E "kernels" | tail -1
done
