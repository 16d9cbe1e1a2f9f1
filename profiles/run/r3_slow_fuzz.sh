#!/bin/bash
# round 3: randomized parity of the ladder form of the slow chaining path (every chunk forced onto it, and the normal mix)
mkdir -p gpurun_out/r3j
SKDER_AMD_FORCE_SLOW=1 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "index_and_triangle or synthetic_with_screen or repeats_indels or structural or repeat_rich or real_derived or benchmark_size or mixed_genome or degenerate or beyond_16 or repetitive_cutoff or dropin or small_batches or low_complexity or truth" > gpurun_out/r3j/pytest_force_slow.log 2>&1
tail -n 4 gpurun_out/r3j/pytest_force_slow.log
t() { timeout $1 python tests/tools/$2 $3 $4 $5 > gpurun_out/r3j/$6.log 2>&1; echo "$6: $(grep -c ' ok' gpurun_out/r3j/$6.log) ok"; grep MISMATCH gpurun_out/r3j/$6.log | head -3; }
export SKDER_AMD_FORCE_SLOW=1
t 150 fuzz_structural.py 7000000 7003000 "" slow_structural
FUZZ_REAL=1 t 150 fuzz_structural.py 7100000 7103000 "" slow_real
t 150 fuzz_repeats.py 7200000 7203000 "" slow_repeats
t 100 fuzz_repeats.py 7300000 7301000 rep slow_rep
unset SKDER_AMD_FORCE_SLOW
t 150 fuzz_structural.py 7400000 7403000 "" structural
FUZZ_REAL=1 t 150 fuzz_structural.py 7500000 7503000 "" real
t 150 fuzz_repeats.py 7600000 7603000 "" repeats
