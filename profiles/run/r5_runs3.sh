#!/bin/bash
# round 5: run loop refill sweep (declined chunks buffered per wavefront), then the per-kernel times of the real-structure workload
export TMPDIR=/tmp
mkdir -p gpurun_out/r5d
for R in 64 48 32 24 16; do
  echo "== refill_min $R"
  SKDER_AMD_RUNS_REFILL=$R D=${D:-8} python profiles/run/r3_real_debug.py 2>&1 | grep -E "^\{" | python -c "
import sys,ast
r=ast.literal_eval(sys.stdin.read().strip().splitlines()[-1])
print({k:(round(v,3) if isinstance(v,float) else v) for k,v in r.items() if k in ('triangle_ms','chain_fast_ms','chain_slow_ms','join_ms','run_extract_ms','finalize_ms','us_per_chained_pair','slow_path_fraction','chained_pairs')})"
done
for R in 64 32; do
cd /tmp && SKDER_AMD_RUNS_REFILL=$R D=8 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r5d/kt$R -o kt --output-format csv -- python3 $GRAFT_REPO_ROOT/profiles/run/r3_real_prof.py > $GRAFT_REPO_ROOT/gpurun_out/r5d/kt$R.log 2>&1
cd $GRAFT_REPO_ROOT
echo "== kernel stats refill_min $R"
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/r5d/kt$R/**/*kernel_stats.csv",recursive=True)
for r in list(csv.DictReader(open(f[0])))[:14]:
    print("%-50s calls %6s avg %10.1f us total %8.2f ms" % (r["Name"].split("(")[0][:50], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
done
