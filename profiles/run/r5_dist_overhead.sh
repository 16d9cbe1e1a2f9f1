#!/bin/bash
# round 5: what the two N > 1 code paths cost on ONE GPU (one rank, RCCL process group) against the plain path
F="--no-cpu-baseline --no-realistic --e2e-genomes 0 --steps 6 --warmup 3"
python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plain ms_per_step', d['ms_per_step']); print(json.dumps(d['roofline'].get('host_wall_ms')))"
for X in replicate components; do
SKDER_AMD_EXCHANGE=$X SKDER_AMD_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py $F 2>gpurun_out/r5_dist_$X.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('forced dist ($X) ms_per_step', d['ms_per_step'], 'edges', d['config']['edges']); print(json.dumps(d['roofline'].get('per_rank_stage_ms'))); print(json.dumps(d['roofline'].get('host_wall_ms'))); print(json.dumps(d.get('exchange')))"
tail -3 gpurun_out/r5_dist_$X.err
done
