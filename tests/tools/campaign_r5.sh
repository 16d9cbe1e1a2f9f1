#!/bin/bash
# round 5: the run loop with lanes that draw new chunks as they finish (chain_runs.hip): fresh seeds, the normal mix of paths, every chunk
# with hits through the run loop (SKDER_AMD_NO_SIEVE=1), and the two ends of the refill threshold (1: a service every round, lanes start
# chunks at any round; 64: a wavefront's chunks start together)
OUT=gpurun_out/fuzz_r5
mkdir -p $OUT
t() { timeout $1 python tests/tools/$2 $3 $4 $5 > $OUT/$6.log 2>&1; echo "$6: $(grep -c ' ok' $OUT/$6.log) ok, $(grep -c MISMATCH $OUT/$6.log) mismatches"; grep MISMATCH $OUT/$6.log | head -3; }
t 90 fuzz_structural.py 11000000 11002500 "" structural
FUZZ_REAL=1 t 90 fuzz_structural.py 11100000 11102500 "" real
t 90 fuzz_repeats.py 11200000 11202500 "" repeats
t 60 fuzz_repeats.py 11300000 11303000 batch batch
SKDER_AMD_NO_SIEVE=1 t 90 fuzz_structural.py 11400000 11402500 "" structural_no_sieve
SKDER_AMD_NO_SIEVE=1 FUZZ_REAL=1 t 90 fuzz_structural.py 11500000 11502500 "" real_no_sieve
SKDER_AMD_NO_SIEVE=1 t 60 fuzz_repeats.py 11600000 11602500 "" repeats_no_sieve
SKDER_AMD_RUNS_REFILL=1 FUZZ_REAL=1 t 60 fuzz_structural.py 11700000 11702500 "" real_refill1
SKDER_AMD_RUNS_REFILL=1 SKDER_AMD_NO_SIEVE=1 t 60 fuzz_repeats.py 11800000 11802500 "" repeats_refill1_no_sieve
SKDER_AMD_RUNS_REFILL=64 FUZZ_REAL=1 t 60 fuzz_structural.py 11900000 11902500 "" real_refill64
