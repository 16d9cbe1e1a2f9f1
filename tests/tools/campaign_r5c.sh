#!/bin/bash
# round 5, third leg: what the other two did not touch directly -- the file-based drop-in after the hybrid inflate path was cut out of the
# ingest (random FASTA formatting, gzip, several members), large genomes, sets appended to
OUT=gpurun_out/fuzz_r5c
mkdir -p $OUT
t() { timeout $1 python tests/tools/$2 $3 $4 $5 > $OUT/$6.log 2>&1; echo "$6: $(grep -c ' ok' $OUT/$6.log) ok, $(grep -c MISMATCH $OUT/$6.log) mismatches"; grep MISMATCH $OUT/$6.log | head -3; }
t 120 fuzz_dropin.py 13000000 13001000 "" dropin
SKDER_AMD_IO_BATCH_MB=1 t 90 fuzz_dropin.py 13100000 13101000 "" dropin_small_batches
t 90 fuzz_repeats.py 13200000 13200600 big big
t 60 fuzz_repeats.py 13300000 13302000 append append
