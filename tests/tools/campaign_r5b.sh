#!/bin/bash
# round 5, second leg: the run loop settles a path whose best end is an EARLIER anchor when at most two anchors lie behind the peak
# (chain_runs.hip, EMIT_PATH / PEAK_OF) -- fresh seeds; the repeat families matter most here (seeds with several occurrences start
# several paths at one seed index: the rule tells them apart by the first anchor's place on the other genome)
OUT=gpurun_out/fuzz_r5b
mkdir -p $OUT
t() { timeout $1 python tests/tools/$2 $3 $4 $5 > $OUT/$6.log 2>&1; echo "$6: $(grep -c ' ok' $OUT/$6.log) ok, $(grep -c MISMATCH $OUT/$6.log) mismatches"; grep MISMATCH $OUT/$6.log | head -3; }
t 100 fuzz_structural.py 12000000 12003000 "" structural
FUZZ_REAL=1 t 100 fuzz_structural.py 12100000 12103000 "" real
t 120 fuzz_repeats.py 12200000 12203000 "" repeats
t 60 fuzz_repeats.py 12250000 12251000 rep rep
t 60 fuzz_repeats.py 12300000 12303000 batch batch
SKDER_AMD_NO_SIEVE=1 t 90 fuzz_structural.py 12400000 12402500 "" structural_no_sieve
SKDER_AMD_NO_SIEVE=1 FUZZ_REAL=1 t 90 fuzz_structural.py 12500000 12502500 "" real_no_sieve
SKDER_AMD_NO_SIEVE=1 t 120 fuzz_repeats.py 12600000 12603000 "" repeats_no_sieve
SKDER_AMD_RUNS_REFILL=1 SKDER_AMD_NO_SIEVE=1 t 60 fuzz_repeats.py 12800000 12802500 "" repeats_refill1_no_sieve
SKDER_AMD_RUNS_REFILL=64 FUZZ_REAL=1 t 60 fuzz_structural.py 12900000 12902500 "" real_refill64
