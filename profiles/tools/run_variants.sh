#!/bin/bash
# usage: profiles/tools/run_variants.sh lib1.so lib2.so ...   (bench each library variant, print kernel times)
cp skder_amd/libskder_amd.so /tmp/lib_orig.so
for L in "$@"; do
  cp exp/$L skder_amd/libskder_amd.so
  echo -n "$L: "
  timeout 120 python bench.py --no-cpu-baseline --steps 2 --warmup 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['roofline']['kernel_ms'].items()}, d['config']['slow_path_chunks'], d['config']['edges'])"
done
cp /tmp/lib_orig.so skder_amd/libskder_amd.so
