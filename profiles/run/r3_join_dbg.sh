#!/bin/bash
# round 3: what the join's time is made of -- the kernel with one of its memory streams switched off at a time (SKDER_AMD_JOIN_DBG:
# every hit word then comes out as "no hit"; durations from rocprofv3's kernel trace, one queue, three steps each)
export TMPDIR=/tmp SKDER_AMD_QUEUES=1
mkdir -p gpurun_out/r3p
for d in ${DBGS:-0 1 2 3 19 32}; do
  rm -rf gpurun_out/r3p/dbg$d
  SKDER_AMD_JOIN_DBG=$d rocprofv3 --kernel-trace -d gpurun_out/r3p/dbg$d -o t -- python3 bench.py --no-realistic --low-mem-genomes 0 --no-cpu-baseline --e2e-genomes 0 --steps 2 --warmup 1 > gpurun_out/r3p/dbg$d.log 2>&1
  echo "dbg=$d $(python3 profiles/run/kstat.py gpurun_out/r3p/dbg$d/*/t_results.db 12 2>/dev/null | grep join_probe || python3 profiles/run/kstat.py gpurun_out/r3p/dbg$d/t_results.db 12 | grep join_probe)"
done
