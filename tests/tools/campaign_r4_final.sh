#!/bin/bash
# round 4, final build: a short confirmation on fresh seeds (the kernels are those of campaign_r4.sh; the run loop's decline list and the
# search path's live mask changed since)
OUT=gpurun_out/fuzz_r4_final
mkdir -p $OUT
t() { timeout $1 python tests/tools/$2 $3 $4 $5 > $OUT/$6.log 2>&1; echo "$6: $(grep -c ' ok' $OUT/$6.log) ok, $(grep -c MISMATCH $OUT/$6.log) mismatches"; grep MISMATCH $OUT/$6.log | head -3; }
t 90 fuzz_structural.py 10000000 10002500 "" structural
FUZZ_REAL=1 t 90 fuzz_structural.py 10100000 10102500 "" real
t 90 fuzz_repeats.py 10200000 10202500 "" repeats
t 60 fuzz_repeats.py 10300000 10303000 batch batch
SKDER_AMD_FORCE_SLOW=1 FUZZ_REAL=1 t 90 fuzz_structural.py 10400000 10402500 "" real_force_rows
t 60 fuzz_dropin.py 10500000 10501000 "" dropin
