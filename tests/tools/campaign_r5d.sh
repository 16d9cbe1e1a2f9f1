#!/bin/bash
# round 5, fourth leg: the round's final build (ring look-back in straight-line form, the sieve's anchor counts added per pair) on fresh seeds
OUT=gpurun_out/fuzz_r5d
mkdir -p $OUT
t() { timeout $1 python tests/tools/$2 $3 $4 $5 > $OUT/$6.log 2>&1; echo "$6: $(grep -c ' ok' $OUT/$6.log) ok, $(grep -c MISMATCH $OUT/$6.log) mismatches"; grep MISMATCH $OUT/$6.log | head -3; }
t 75 fuzz_structural.py 14000000 14003000 "" structural
FUZZ_REAL=1 t 75 fuzz_structural.py 14100000 14103000 "" real
t 75 fuzz_repeats.py 14200000 14203000 "" repeats
SKDER_AMD_NO_SIEVE=1 FUZZ_REAL=1 t 60 fuzz_structural.py 14500000 14502500 "" real_no_sieve
SKDER_AMD_NO_SIEVE=1 t 60 fuzz_repeats.py 14600000 14603000 "" repeats_no_sieve
t 45 fuzz_repeats.py 14300000 14303000 batch batch
