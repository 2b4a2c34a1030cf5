#!/usr/bin/env bash
# round 6, call D: residency of the step kernel's waves (HW_ID + first / last cycle stamp of every wave, tools/phase_timing.py)
set -u
OUT=gpurun_out/r06_d
mkdir -p $OUT
export JAXSIM_AMD_SPEC_EXTRA_FLAGS=-DJXS_PHASE_TIMING
for n in 1024 8192 65536; do
  JAXSIM_AMD_SPECIALIZE=1 python tools/phase_timing.py $n > $OUT/phases_$n.log 2>&1
  echo "== N=$n"; grep -E "total|placement|residency|first start" $OUT/phases_$n.log
done
