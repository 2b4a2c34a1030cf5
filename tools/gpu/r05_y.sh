#!/usr/bin/env bash
# round 5, call Y: the three device fuzz campaigns on the final sources (tools/fuzz/gpu_campaign.py seeds 31 and 37,
# gpu_campaign_queries.py seed 47; cases prepared on the CPU)
set -u
OUT=gpurun_out/r05_y
mkdir -p $OUT
timeout 900 python tools/fuzz/gpu_campaign.py run tools/fuzz/_cases.pkl $OUT/gpu_fuzz_campaign.txt > $OUT/run1.log 2>&1; echo "campaign 1 rc=$?"
timeout 900 python tools/fuzz/gpu_campaign.py run tools/fuzz/_cases2.pkl $OUT/gpu_fuzz_campaign2.txt > $OUT/run2.log 2>&1; echo "campaign 2 rc=$?"
timeout 900 python tools/fuzz/gpu_campaign_queries.py run tools/fuzz/_cases_q.pkl $OUT/gpu_fuzz_queries.txt > $OUT/run3.log 2>&1; echo "queries rc=$?"
grep -h "campaign seed" $OUT/*.txt | cut -c1-400
