#!/usr/bin/env bash
# round 5, call T: the fuzz campaign on the device (tools/fuzz/gpu_campaign.py; cases, truth and emulation results prepared on the CPU)
set -u
OUT=gpurun_out/r05_t
mkdir -p $OUT
timeout 1200 python tools/fuzz/gpu_campaign.py run tools/fuzz/_cases.pkl $OUT/gpu_campaign.txt > $OUT/run.log 2> $OUT/err.log
echo "rc=$?"
tail -30 $OUT/run.log
tail -5 $OUT/err.log
