#!/bin/bash
# round 5: per-kernel times of the real-structure workload (34 assemblies x D descendants), rocprofv3 --kernel-trace --stats
export TMPDIR=/tmp
TAG=${TAG:-r5}
mkdir -p gpurun_out/r5d
D=${D:-8} rocprofv3 --kernel-trace --stats -d gpurun_out/r5d/kt_$TAG -o kt --output-format csv -- python3 profiles/run/r3_real_prof.py > gpurun_out/r5d/kt_$TAG.log 2>&1
tail -1 gpurun_out/r5d/kt_$TAG.log | cut -c1-400
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/r5d/kt_$TAG/**/*kernel_stats.csv",recursive=True)
rows=list(csv.DictReader(open(f[0])))
open("gpurun_out/r5d/kernel_stats_$TAG.csv","w").write(open(f[0]).read())
for r in rows[:16]:
    print("%-50s calls %6s avg %10.1f us total %8.2f ms" % (r["Name"].split("(")[0][:50], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
