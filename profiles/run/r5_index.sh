#!/bin/bash
# round 5: index build A/B -- with the packed (k-mer, position, record) copy of the seeds and without it
export TMPDIR=/tmp
for V in 1 0; do
  (cd skder_amd/csrc && touch index.hip && make EXTRA=-DIDX_PACKED=$V 2>&1 | grep -E "error")
  echo "== IDX_PACKED=$V"
  python profiles/run/r5_index.py
done
