#!/bin/bash
# round 4: the two at-size runs on one MI355X: config 5 at 50,000 genomes (streamed sketch stage), low_mem_greedy on 20,000 genomes of one species
mkdir -p gpurun_out/r4f
timeout 900 python profiles/run/r4_scale_50000.py > gpurun_out/r4f/scale_50000.json 2> gpurun_out/r4f/scale_50000.err; echo "scale rc $?"; tail -c 2500 gpurun_out/r4f/scale_50000.json; tail -n 3 gpurun_out/r4f/scale_50000.err
N=20000 timeout 1500 python profiles/run/r4_one_species.py > gpurun_out/r4f/one_species_20000.json 2> gpurun_out/r4f/one_species_20000.err; echo "one species rc $?"; tail -c 1800 gpurun_out/r4f/one_species_20000.json; tail -n 3 gpurun_out/r4f/one_species_20000.err
