#!/bin/bash
# round 5, after finalize went to 192 threads per pair with its LDS sized to the batch: the main modes on fresh seeds (8 GPU-minutes);
# campaign_r5g.sh (big, append, queues, refill thresholds) was repeated on this build as well
OUT=gpurun_out/fuzz_r5h
mkdir -p $OUT
t() { timeout $1 python tests/tools/$2 $3 $4 $5 > $OUT/$6.log 2>&1; echo "$6: $(grep -c ' ok' $OUT/$6.log) ok, $(grep -c MISMATCH $OUT/$6.log) mismatches"; grep MISMATCH $OUT/$6.log | head -3; }
t 120 fuzz_structural.py 18000000 18009000 "" structural
FUZZ_REAL=1 t 120 fuzz_structural.py 18100000 18109000 "" real
t 120 fuzz_repeats.py 18200000 18209000 "" repeats
t 60 fuzz_repeats.py 18250000 18251000 rep rep
SKDER_AMD_FORCE_SLOW=1 FUZZ_REAL=1 t 60 fuzz_structural.py 18700000 18703000 "" real_force_rows
