#!/usr/bin/env bash
# round 6, call P: the query / rollout campaign on the device (tools/fuzz/gpu_campaign_queries.py, seed 71, 600 trees) -- now with
# js.ode.system_dynamics and js.contact.link_contact_forces for the three contact models (a quarter on height fields) and hubs
set -u
OUT=gpurun_out/r06_p
mkdir -p $OUT
timeout 2400 python tools/fuzz/gpu_campaign_queries.py run tools/fuzz/_cases_q6.pkl $OUT/gpu_queries.txt > $OUT/run.log 2> $OUT/err.log
echo "rc=$?"
tail -60 $OUT/run.log | cut -c1-200
tail -5 $OUT/err.log
