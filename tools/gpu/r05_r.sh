#!/usr/bin/env bash
# round 5, call R: long runs of the tree solve, and a clean bench line at --steps 20
set -u
OUT=gpurun_out/r05_r
mkdir -p $OUT
export JAXSIM_AMD_SPECIALIZE=1
timeout 900 python tools/relaxed_long_run.py 2000 4096 > $OUT/long_run.txt 2>&1; cat $OUT/long_run.txt | grep -v Warning
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> $OUT/bench.err
python -c "import json; d=json.load(open('$OUT/bench_steps20.json')); print('value', round(d['value']/1e6,2), d['ms_per_step']*1e3, d['roofline']['frac'], d['roofline']['kernel_avg_launch_us'])"
