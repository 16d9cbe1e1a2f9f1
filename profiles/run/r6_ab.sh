#!/bin/bash
# profiles/run/r6_ab.sh V1 V2 ... -- A/B on one box: builds of the library kept as skder_amd/lib_<V>.so.bin (git-ignored, they travel with
# the snapshot), the same short bench command for each, interleaved, REPS times; prints ms per step and the per-kernel HIP-event figures.
# The last variant named stays installed.
REPS=${REPS:-2}
OUT=${OUT:-gpurun_out/r6/ab}
mkdir -p $OUT
B="--steps ${STEPS:-10} --warmup 3 --no-cpu-baseline --no-realistic --e2e-genomes 0 --parity-pairs 0 --low-mem-genomes 0 --one-species-genomes 0 ${EXTRA_ARGS:-}"
for rep in $(seq 1 $REPS); do for v in "$@"; do
  cp skder_amd/lib_$v.so.bin skder_amd/libskder_amd.so
  python bench.py $B 2>$OUT/err_${v}_$rep.log | tail -1 > $OUT/line_${v}_$rep.json
  python - "$v" $OUT/line_${v}_$rep.json <<'PY'
import json, sys
v, f = sys.argv[1], sys.argv[2]
try:
    d = json.load(open(f)); r = d["roofline"]
    km = {k.split("_kernel")[0][:14]: round(x, 2) for k, x in r["kernel_ms"].items()}
    om = {k[:12]: round(x, 2) for k, x in r["other_ms"].items()}
    print("%-8s %.2f ms/step  1q %.2f  %s %s edges %d" % (v, d["ms_per_step"], r["ms_per_step_one_queue"] or 0, km, om, d["config"]["edges"]))
except Exception as ex:
    print(v, "FAILED", ex, open(f.replace("line_", "err_").replace(".json", ".log")).read()[-800:])
PY
done; done
