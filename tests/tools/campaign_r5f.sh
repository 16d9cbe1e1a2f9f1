#!/bin/bash
# round 5: after the slot-major chain slots and the finalize changes (marks in LDS, 256 bins), fresh seeds
OUT=gpurun_out/fuzz_r5f
mkdir -p $OUT
t() { timeout $1 python tests/tools/$2 $3 $4 $5 > $OUT/$6.log 2>&1; echo "$6: $(grep -c ' ok' $OUT/$6.log) ok, $(grep -c MISMATCH $OUT/$6.log) mismatches"; grep MISMATCH $OUT/$6.log | head -3; }
t 240 fuzz_structural.py 16000000 16009000 "" structural
FUZZ_REAL=1 t 240 fuzz_structural.py 16100000 16109000 "" real
t 240 fuzz_repeats.py 16200000 16209000 "" repeats
t 90 fuzz_repeats.py 16250000 16251000 rep rep
t 120 fuzz_repeats.py 16300000 16309000 batch batch
SKDER_AMD_NO_SIEVE=1 FUZZ_REAL=1 t 180 fuzz_structural.py 16500000 16509000 "" real_no_sieve
SKDER_AMD_NO_SIEVE=1 t 180 fuzz_repeats.py 16600000 16609000 "" repeats_no_sieve
SKDER_AMD_FORCE_SLOW=1 FUZZ_REAL=1 t 90 fuzz_structural.py 16700000 16703000 "" real_force_rows
t 90 fuzz_dropin.py 16800000 16801000 "" dropin
