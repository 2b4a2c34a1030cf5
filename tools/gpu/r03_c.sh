#!/usr/bin/env bash
# round 3, GPU call C: two-wave variant v2 (kinematics handed over, base factor by the inertia wave, even placement)
set -u
R=$PWD
OUT=$R/gpurun_out/r03_e
mkdir -p "$OUT"
for cfg in "JXS_DUO=0" "JXS_DUO=1" "JXS_DUO=1 JXS_DUO_MAX_BLOCKS=4096"; do
  env $cfg timeout 300 python tools/sweep.py --sizes 1024,2048,4096 --steps 1000 2>&1 | sed "s/^/$cfg /" | tee -a "$OUT/summary.txt"
done
export JAXSIM_AMD_SPEC_EXTRA_FLAGS=-DJXS_PHASE_TIMING
JXS_DUO=1 JAXSIM_AMD_SPECIALIZE=1 timeout 300 python tools/phase_timing.py > "$OUT/phases_duo.log" 2>&1
JXS_DUO=0 JAXSIM_AMD_SPECIALIZE=1 timeout 300 python tools/phase_timing.py > "$OUT/phases_solo.log" 2>&1
unset JAXSIM_AMD_SPEC_EXTRA_FLAGS
cat "$OUT/phases_duo.log"; cat "$OUT/phases_solo.log"
timeout 900 python -m pytest tests -m gpu -q -x > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"
tail -5 "$OUT/pytest.log"
