#!/usr/bin/env bash
# round 3: first-load latency of the step kernel against the number of waves in flight (is the prologue latency or a burst?)
set -u
export JAXSIM_AMD_SPEC_EXTRA_FLAGS="-DJXS_PHASE_TIMING"
for n in 2 64 256 1024 4096; do
  JXS_DUO=0 JAXSIM_AMD_SPECIALIZE=1 timeout 300 python tools/phase_timing.py $n 2>&1 | grep -E "^N=|loads arrive|total|index tables|base state|point tables|joint state"
done
