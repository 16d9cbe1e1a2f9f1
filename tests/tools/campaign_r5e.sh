#!/bin/bash
# round 5, last leg: a longer run of the round's final build on fresh seeds (25 GPU-minutes)
OUT=gpurun_out/fuzz_r5e
mkdir -p $OUT
t() { timeout $1 python tests/tools/$2 $3 $4 $5 > $OUT/$6.log 2>&1; echo "$6: $(grep -c ' ok' $OUT/$6.log) ok, $(grep -c MISMATCH $OUT/$6.log) mismatches"; grep MISMATCH $OUT/$6.log | head -3; }
t 240 fuzz_structural.py 15000000 15009000 "" structural
FUZZ_REAL=1 t 240 fuzz_structural.py 15100000 15109000 "" real
t 240 fuzz_repeats.py 15200000 15209000 "" repeats
t 90 fuzz_repeats.py 15250000 15251000 rep rep
t 120 fuzz_repeats.py 15300000 15309000 batch batch
SKDER_AMD_NO_SIEVE=1 FUZZ_REAL=1 t 180 fuzz_structural.py 15500000 15509000 "" real_no_sieve
SKDER_AMD_NO_SIEVE=1 t 180 fuzz_repeats.py 15600000 15609000 "" repeats_no_sieve
SKDER_AMD_FORCE_SLOW=1 FUZZ_REAL=1 t 90 fuzz_structural.py 15700000 15703000 "" real_force_rows
t 90 fuzz_dropin.py 15800000 15801000 "" dropin
