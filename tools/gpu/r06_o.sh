#!/usr/bin/env bash
# round 6, call O: the fuzz campaign on the device on the round's final sources (tools/fuzz/gpu_campaign.py: 1500 random trees
# incl. hubs with 7 - 12 legs and height-field terrains; cases, truth and emulation results prepared on the CPU, seed 61)
set -u
OUT=gpurun_out/r06_o
mkdir -p $OUT
timeout 2400 python tools/fuzz/gpu_campaign.py run tools/fuzz/_cases_r6.pkl $OUT/gpu_campaign.txt > $OUT/run.log 2> $OUT/err.log
echo "rc=$?"
tail -40 $OUT/run.log
tail -5 $OUT/err.log
