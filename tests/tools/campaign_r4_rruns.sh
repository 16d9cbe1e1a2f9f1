#!/bin/bash
# round 4, the run DP in row form (chain_rruns.hip, opt-in): its parity test, then the randomized families with it behind the run loop
# (SKDER_AMD_RRUNS=1) and instead of it (=2; with SKDER_AMD_NO_SIEVE=1 every chunk with hits goes through it)
OUT=gpurun_out/fuzz_r4_rruns
mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "run_dp_in_row_form" > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest.log
t() { timeout $1 python tests/tools/$2 $3 $4 $5 > $OUT/$6.log 2>&1; echo "$6: $(grep -c ' ok' $OUT/$6.log) ok, $(grep -c MISMATCH $OUT/$6.log) mismatches"; grep MISMATCH $OUT/$6.log | head -3; }
SKDER_AMD_RRUNS=1 t 150 fuzz_structural.py 9200000 9202500 "" structural_rr1
SKDER_AMD_RRUNS=1 FUZZ_REAL=1 t 150 fuzz_structural.py 9300000 9302500 "" real_rr1
SKDER_AMD_RRUNS=2 t 150 fuzz_structural.py 9400000 9402500 "" structural_rr2
SKDER_AMD_RRUNS=2 FUZZ_REAL=1 t 150 fuzz_structural.py 9500000 9502500 "" real_rr2
SKDER_AMD_RRUNS=2 t 150 fuzz_repeats.py 9600000 9602500 "" repeats_rr2
SKDER_AMD_RRUNS=2 t 100 fuzz_repeats.py 9700000 9701000 rep rep_rr2
SKDER_AMD_RRUNS=2 SKDER_AMD_NO_SIEVE=1 t 150 fuzz_structural.py 9800000 9802500 "" structural_rr2_nosieve
SKDER_AMD_RRUNS=2 SKDER_AMD_NO_SIEVE=1 t 150 fuzz_repeats.py 9900000 9902500 "" repeats_rr2_nosieve
