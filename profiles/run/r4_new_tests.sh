#!/bin/bash
# round 4: the new GPU tests, the fixture dump of the 1,000-genome table, the one-species low_mem_greedy leg at 5,000 genomes
mkdir -p gpurun_out/r4e
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "device_descendants or tc_grid or config5_shape or several_gpus" > gpurun_out/r4e/pytest_new.log 2>&1; echo "new tests rc $?"
tail -n 5 gpurun_out/r4e/pytest_new.log
SKDER_AMD_DUMP_N1000=gpurun_out/n1000 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "1000_genomes" > gpurun_out/r4e/pytest_n1000.log 2>&1; echo "n1000 rc $?"
tail -n 3 gpurun_out/r4e/pytest_n1000.log
gzip -1 gpurun_out/n1000/table.tsv
N=5000 python profiles/run/r4_one_species.py > gpurun_out/r4e/one_species_5000.json 2> gpurun_out/r4e/one_species_5000.err; tail -c 1500 gpurun_out/r4e/one_species_5000.json; tail -n 3 gpurun_out/r4e/one_species_5000.err
