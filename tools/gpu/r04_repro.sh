#!/usr/bin/env bash
# round 4: the round-3 NaN reproducer over kernel-source variants (tools/ab/<variant>), fresh processes
set -u
R=$PWD
OUT=$R/gpurun_out/${1:-r04_repro}
shift
mkdir -p "$OUT"
export JAXSIM_AMD_SPECIALIZE=cached
for i in 1 2 3 4; do
  for v in "$@"; do
    if [ "$v" = cur ]; then timeout 120 python tools/ab/repro.py run ${DT:-float32} 2>&1 | tail -1 | tee -a "$OUT/summary.txt"
    else JAXSIM_AMD_SPEC_CSRC=/root/repo/tools/ab/$v timeout 120 python tools/ab/repro.py run ${DT:-float32} 2>&1 | tail -1 | tee -a "$OUT/summary.txt"; fi
  done
done
