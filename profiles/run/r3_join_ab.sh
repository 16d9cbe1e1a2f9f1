#!/bin/bash
# round 3: the join's probes side by side -- round 2's per-entry probe (SKDER_AMD_JOIN_V1), the packed probe with one and with two
# sub-trips per trip; rocprofv3 kernel trace, one queue, 3 steps each, same box
export TMPDIR=/tmp SKDER_AMD_QUEUES=1
mkdir -p gpurun_out/r3p
for v in V1 SUB1 SUB2 SUB1 SUB2 V1; do
  unset SKDER_AMD_JOIN_V1 SKDER_AMD_JOIN_SUB
  case $v in V1) export SKDER_AMD_JOIN_V1=1;; SUB1) export SKDER_AMD_JOIN_SUB=1;; SUB2) export SKDER_AMD_JOIN_SUB=2;; esac
  rm -rf gpurun_out/r3p/ab
  rocprofv3 --kernel-trace -d gpurun_out/r3p/ab -o t -- python3 bench.py --no-realistic --low-mem-genomes 0 --no-cpu-baseline --e2e-genomes 0 --steps 2 --warmup 1 > gpurun_out/r3p/ab.log 2>&1
  echo "$v $(python3 profiles/run/kstat.py gpurun_out/r3p/ab/*/t_results.db 12 2>/dev/null | grep join_probe || python3 profiles/run/kstat.py gpurun_out/r3p/ab/t_results.db 12 | grep join_probe)"
done
