#!/usr/bin/env bash
# round 3, GPU call Q: step kernel after the scalar-base addressing / identity padding lane / zero-lane pulls
set -u
R=$PWD
OUT=$R/gpurun_out/r03_r
mkdir -p "$OUT"
timeout 900 python -m pytest tests -m gpu -q -x > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"
tail -3 "$OUT/pytest.log"
for i in 1 2; do
JXS_DUO=0 timeout 300 python tools/sweep.py --sizes 1024,2048,65536 --steps 1000 2>&1 | tee -a "$OUT/summary.txt"
done
JXS_DUO=1 timeout 300 python tools/sweep.py --sizes 1024 --steps 1000 2>&1 | sed "s/^/JXS_DUO=1 /" | tee -a "$OUT/summary.txt"
JAXSIM_AMD_SPECIALIZE=0 timeout 300 python tools/sweep.py --sizes 1024 --steps 1000 2>&1 | sed "s/^/generic /" | tee -a "$OUT/summary.txt"
export JAXSIM_AMD_SPEC_EXTRA_FLAGS=-DJXS_PHASE_TIMING
JXS_DUO=0 JAXSIM_AMD_SPECIALIZE=1 timeout 300 python tools/phase_timing.py 2>&1 | head -20
