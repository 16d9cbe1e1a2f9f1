#!/bin/bash
# round 4: what the N > 1 code path costs on ONE GPU (one rank, RCCL process group) against the plain path -- VERDICT r3 item 6's gate (<= 6 ms; r3: 13.8)
F="--no-cpu-baseline --no-realistic --e2e-genomes 0 --steps 6 --warmup 3"
python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plain ms_per_step', d['ms_per_step']); print(json.dumps(d['roofline'].get('host_wall_ms')))"
SKDER_AMD_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('forced dist ms_per_step', d['ms_per_step']); print(json.dumps(d['roofline'].get('per_rank_stage_ms'))); print(json.dumps(d['roofline'].get('host_wall_ms')))"
