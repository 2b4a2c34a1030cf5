#!/usr/bin/env bash
# round 4, call O: fp64 sincos A/B in one call (model-specialised step kernel, fp64 headline model)
set -u
export JAXSIM_AMD_SPECIALIZE=1
for rep in 1 2 3; do
  for f in "" "-DJXS_LIBM_SINCOS64"; do
    export JAXSIM_AMD_SPEC_EXTRA_FLAGS=$f
    python bench.py --gpus 1 --dtype float64 --steps 200 --warmup 20 --no-cpu-baseline --saturated-envs 0 --no-other-contact-models 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$f] fp64 step', round(d['ms_per_step']*1e3,3), 'us; kernel', round(d['roofline']['kernel_avg_launch_us'],3))"
  done
done
