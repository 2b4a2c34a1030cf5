#!/usr/bin/env bash
# round 5, call O: config 5 with the interior-point iteration warm-started (KParams::qp_warm) against CVXGEN's initial point
set -u
OUT=gpurun_out/r05_o
mkdir -p $OUT
export JAXSIM_AMD_SPECIALIZE=1
for rep in 1 2 3; do
  for w in 1 0; do
    if [ $w = 0 ]; then export JXS_DISABLE_QP_WARM=1; else unset JXS_DISABLE_QP_WARM; fi
    for st in "" "--standing"; do
      timeout 300 python tools/bench_c5.py --points 4 $st 2>> $OUT/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('warm=$w $st', round(d['ms_per_step']*1e3,2), 'us', round(d['env_steps_per_s']/1e6,2), 'M/s finite', d['finite_envs'])" | tee -a $OUT/ab.txt
    done
  done
done
unset JXS_DISABLE_QP_WARM
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --saturated-envs 0 --no-python-loop > $OUT/bench.json 2> $OUT/bench.err
python -c "import json; d=json.load(open('$OUT/bench.json')); print(json.dumps(d['other_contact_models']['config5_rigid_contacts'])[:700])"
JAXSIM_AMD_SPEC_EXTRA_FLAGS=-DJXS_PHASE_TIMING python tools/phase_timing_rigid.py 4 4096 > $OUT/phases.txt 2>&1; cat $OUT/phases.txt
