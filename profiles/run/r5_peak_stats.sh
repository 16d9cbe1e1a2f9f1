#!/bin/bash
# round 5: what is left of "best end is not the last anchor" after the short tails were settled: a measurement build that files the peak's
# run not being in the ring under qrep and a tail of three or more anchors under records-full (the run loop uses neither of those two causes)
(cd skder_amd/csrc && touch chain_runs.hip && make EXTRA=-DSKDER_PEAK_STATS 2>&1 | grep -E "error")
D=8 python profiles/run/r3_real_debug.py 2>&1 | grep -E "batch:" | tail -2
