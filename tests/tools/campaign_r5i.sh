#!/bin/bash
# round 5, after the finalize retry asks for what the fullest pair wants: the modes whose pairs run over the LDS estimate (4 GPU-minutes)
OUT=gpurun_out/fuzz_r5i
mkdir -p $OUT
t() { timeout $1 python tests/tools/$2 $3 $4 $5 > $OUT/$6.log 2>&1; echo "$6: $(grep -c ' ok' $OUT/$6.log) ok, $(grep -c MISMATCH $OUT/$6.log) mismatches"; grep MISMATCH $OUT/$6.log | head -3; }
t 70 fuzz_repeats.py 19000000 19009000 rep rep
t 70 fuzz_repeats.py 19100000 19109000 "" repeats
FUZZ_REAL=1 t 60 fuzz_structural.py 19200000 19209000 "" real
t 40 fuzz_repeats.py 19300000 19309000 big big
