#!/usr/bin/env bash
# round 5, call A: (1) the chain hand-off micro-benchmark (tools/ubench/chain_handoff.hip): can launch k + 1 start and
# load its tables under launch k, with a per-tile flag for the state?  (2) the lane-group A/B VERDICT r4 asked for:
# the headline step with 64 lanes per environment (one environment per wave, 1024 waves) against the default 32.
set -u
OUT=gpurun_out/r05_a
mkdir -p $OUT
for w in 450 300; do
  timeout 300 tools/ubench/chain_handoff 512 2000 $w > $OUT/chain_512_w$w.txt 2>&1
done
timeout 300 tools/ubench/chain_handoff 1024 2000 450 > $OUT/chain_1024_w450.txt 2>&1
tail -50 $OUT/chain_512_w450.txt
export JAXSIM_AMD_SPECIALIZE=1
B="python bench.py --gpus 1 --steps 2000 --warmup 20 --no-cpu-baseline --saturated-envs 0 --no-other-contact-models"
for rep in 1 2; do
  for g in 0 64; do
    if [ $g = 64 ]; then export JXS_MIN_LANES=64; else unset JXS_MIN_LANES; fi
    timeout 600 $B 2> $OUT/bench_g${g}_$rep.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('min_lanes=$g', round(d['value']/1e6,2), 'M env-steps/s', round(d['ms_per_step']*1e3,3), 'us', d['config'].get('kernel'))" | tee -a $OUT/lanes_ab.txt
  done
done
