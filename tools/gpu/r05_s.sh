#!/usr/bin/env bash
# round 5, call S: compiler scheduling flags on the model-specialised step kernel (no source change): steady-state step time
set -u
OUT=gpurun_out/r05_s
mkdir -p $OUT
export JAXSIM_AMD_SPECIALIZE=1
B="python bench.py --gpus 1 --steps 2000 --warmup 50 --no-cpu-baseline --saturated-envs 0 --no-other-contact-models --no-python-loop"
for rep in 1 2; do
  for f in "" "-mllvm -amdgpu-schedule-relaxed-occupancy=true" "-mllvm -amdgpu-schedule-metric-bias=0" "-mllvm -misched-postra=1" "-mllvm -enable-post-misched=0"; do
    JAXSIM_AMD_SPEC_EXTRA_FLAGS="$f" timeout 300 $B 2>> $OUT/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('flags=[$f]', round(d['ms_per_step']*1e3,3), 'us', round(d['value']/1e6,2), 'M', d['roofline'].get('valu_per_wave_static'))" | tee -a $OUT/ab.txt
  done
done
