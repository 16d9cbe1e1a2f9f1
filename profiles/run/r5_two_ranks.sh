#!/bin/bash
# round 5: bench.py with two gloo ranks sharing the one GPU (functional check of the N > 1 line, both exchanges priced in one run)
F="--genomes 400 --genome-len 1000000 --steps 3 --warmup 1 --no-cpu-baseline"
SKDER_AMD_OTHER_EXCHANGE=1 SKDER_AMD_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 $F 2>gpurun_out/r5_two_ranks.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('n_gpus', d['n_gpus'], 'ms', d['ms_per_step'], 'edges', d['config']['edges'])
print(json.dumps(d['exchange']))
print(json.dumps(d['roofline']['per_rank_stage_ms']))"
tail -2 gpurun_out/r5_two_ranks.err
