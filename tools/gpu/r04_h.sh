#!/usr/bin/env bash
# round 4, call H: register budget of the step kernel (waves per SIMD) against the saturated figure
set -u
R=$PWD; OUT=$R/gpurun_out/r04_h; mkdir -p "$OUT"
for w in 1 4 5; do
  export JAXSIM_AMD_SPEC_EXTRA_FLAGS=-DJXS_MIN_WAVES=$w
  JAXSIM_AMD_SPECIALIZE=1 timeout 600 python tools/sweep.py --sizes 1024,4096,16384,65536,262144 --steps 300 2>&1 | grep "N=" | sed "s/^/min_waves=$w /"
done
