#!/usr/bin/env bash
# round 3: A/B of the working tree's step kernel against the kernel sources kept in tools/ab/prev (same library, same box)
set -u
R=$PWD
OUT=$R/gpurun_out/${1:-r03_ab}
mkdir -p "$OUT"
for i in 1 2 3; do
  JAXSIM_AMD_SPECIALIZE=1 timeout 300 python tools/sweep.py --sizes 1024,2048 --steps 1000 2>&1 | sed "s/^/new:  /" | tee -a "$OUT/summary.txt"
  JAXSIM_AMD_SPEC_CSRC=$R/tools/ab/prev JAXSIM_AMD_SPECIALIZE=1 timeout 300 python tools/sweep.py --sizes 1024,2048 --steps 1000 2>&1 | sed "s/^/prev: /" | tee -a "$OUT/summary.txt"
done
