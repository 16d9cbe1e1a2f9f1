#!/bin/bash
# measurement build (-DFIN_TIMING): where a finalize workgroup spends its time, headline and real-structure workloads
cp skder_amd/libskder_amd.so /tmp/keep.so; cp skder_amd/lib_timing.so.bin skder_amd/libskder_amd.so
B="--steps 2 --warmup 1 --no-cpu-baseline --no-realistic --e2e-genomes 0 --parity-pairs 0 --low-mem-genomes 0 --one-species-genomes 0"
SKDER_AMD_DEBUG=1 python bench.py $B 2>&1 | grep "finalize ticks" | python -c "
import sys,re
t=[0]*6
for l in sys.stdin:
    v=[int(x) for x in re.findall(r'(\d+)[,;]', l)] + [int(re.findall(r'workgroups (\d+)', l)[0])]
    t=[a+b for a,b in zip(t,v)]
print('headline: per workgroup us: gather %.2f bins %.2f rounds %.2f sums %.2f roots %.2f (workgroups %d)' % tuple([x/100.0/t[5] for x in t[:5]]+[t[5]]))
"
D=30 python profiles/run/r3_real_debug.py 2>&1 | grep "finalize ticks" | python -c "
import sys,re
t=[0]*6
for l in sys.stdin:
    v=[int(x) for x in re.findall(r'(\d+)[,;]', l)] + [int(re.findall(r'workgroups (\d+)', l)[0])]
    t=[a+b for a,b in zip(t,v)]
print('real: per workgroup us: gather %.2f bins %.2f rounds %.2f sums %.2f roots %.2f (workgroups %d)' % tuple([x/100.0/t[5] for x in t[:5]]+[t[5]]))
"
cp /tmp/keep.so skder_amd/libskder_amd.so
