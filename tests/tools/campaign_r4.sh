#!/bin/bash
# round 4 campaign: the row kernel (chain_rows.hip) as the general path.  Full GPU suite, then the parity tests and the randomized
# families with every chunk forced onto the row kernel (SKDER_AMD_FORCE_SLOW=1), with the row kernel switched off (everything it
# would take goes to the one-wavefront-per-chunk kernel: SKDER_AMD_NO_ROWS=1), and with the sieve off (SKDER_AMD_NO_SIEVE=1)
B=${B:-0}            # seed offset: B=30000000 runs the same tools on fresh families
OUT=${OUT:-gpurun_out/fuzz_r4}
mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?"
SEL="index_and_triangle or synthetic_with_screen or repeats_indels or repeat_rich or benchmark_size or mixed_genome or degenerate or beyond_16 or repetitive_cutoff or dropin"
SKDER_AMD_FORCE_SLOW=1 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "$SEL" > $OUT/pytest_force_rows.log 2>&1; echo "force rows rc $?"
SKDER_AMD_NO_ROWS=1 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "$SEL or structural or real_derived" > $OUT/pytest_no_rows.log 2>&1; echo "no rows rc $?"
SKDER_AMD_NO_SIEVE=1 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "$SEL or structural or real_derived" > $OUT/pytest_no_sieve.log 2>&1; echo "no sieve rc $?"
t() { timeout $1 python tests/tools/$2 $3 $4 $5 > $OUT/$6.log 2>&1; echo "$6: $(grep -c ' ok' $OUT/$6.log) ok, $(grep -c MISMATCH $OUT/$6.log) mismatches"; grep MISMATCH $OUT/$6.log | head -3; }
t 200 fuzz_structural.py $((B+8000000)) $((B+8002500)) "" structural
FUZZ_REAL=1 t 200 fuzz_structural.py $((B+8100000)) $((B+8102500)) "" real
t 200 fuzz_repeats.py $((B+8200000)) $((B+8202500)) "" repeats
t 100 fuzz_repeats.py $((B+8300000)) $((B+8301000)) rep rep
t 200 fuzz_repeats.py $((B+8400000)) $((B+8403000)) batch batch
SKDER_AMD_FORCE_SLOW=1 t 200 fuzz_structural.py $((B+8500000)) $((B+8502500)) "" structural_force_rows
SKDER_AMD_FORCE_SLOW=1 FUZZ_REAL=1 t 200 fuzz_structural.py $((B+8600000)) $((B+8602500)) "" real_force_rows
SKDER_AMD_FORCE_SLOW=1 t 200 fuzz_repeats.py $((B+8700000)) $((B+8702500)) "" repeats_force_rows
SKDER_AMD_FORCE_SLOW=1 t 100 fuzz_repeats.py $((B+8800000)) $((B+8801000)) rep rep_force_rows
SKDER_AMD_FORCE_SLOW=1 t 100 fuzz_repeats.py $((B+8900000)) $((B+8901000)) big big_force_rows
SKDER_AMD_NO_ROWS=1 t 100 fuzz_structural.py $((B+9000000)) $((B+9001000)) "" structural_no_rows
t 100 fuzz_dropin.py $((B+9100000)) $((B+9101000)) "" dropin
