#!/bin/bash
# round 3: the judged profile set (kernel trace as shipped and on one queue, FETCH/WRITE passes + calibration, SQ counters, bench line)
bash profiles/collect.sh round3 > gpurun_out/collect_round3.log 2>&1
tail -5 gpurun_out/collect_round3.log
python -m pytest tests/ -x -q -m gpu > gpurun_out/pytest_round3.log 2>&1; tail -3 gpurun_out/pytest_round3.log
