#!/bin/bash
# round 5, the modes the sixth leg left out, on the round's final build and fresh seeds: big genomes, appended sets, small batches on two and
# three queues, the two ends of the run loop's refill threshold (15 GPU-minutes)
OUT=gpurun_out/fuzz_r5g
mkdir -p $OUT
t() { timeout $1 python tests/tools/$2 $3 $4 $5 > $OUT/$6.log 2>&1; echo "$6: $(grep -c ' ok' $OUT/$6.log) ok, $(grep -c MISMATCH $OUT/$6.log) mismatches"; grep MISMATCH $OUT/$6.log | head -3; }
t 180 fuzz_repeats.py 17000000 17009000 big big
t 120 fuzz_repeats.py 17100000 17109000 append append
SKDER_AMD_QUEUES=2 t 120 fuzz_repeats.py 17200000 17209000 batch batch_two_queues
SKDER_AMD_QUEUES=3 t 120 fuzz_repeats.py 17300000 17309000 batch batch_three_queues
SKDER_AMD_RUNS_REFILL=1 FUZZ_REAL=1 t 120 fuzz_structural.py 17400000 17409000 "" real_refill_1
SKDER_AMD_RUNS_REFILL=64 FUZZ_REAL=1 t 120 fuzz_structural.py 17500000 17509000 "" real_refill_64
SKDER_AMD_RUNS_REFILL=1 t 90 fuzz_repeats.py 17600000 17609000 "" repeats_refill_1
