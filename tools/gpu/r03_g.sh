#!/usr/bin/env bash
# round 3, GPU call G: config 5 after the DPP-broadcast small solver
set -u
R=$PWD
OUT=$R/gpurun_out/r03_h
mkdir -p "$OUT"
timeout 900 python -m pytest tests -m gpu -q -x > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"
tail -4 "$OUT/pytest.log"
for a in "" "--standing" "--envs 16384" "--points 16" "--contact relaxed" "--contact relaxed --points 16" "--contact relaxed --points 32 --envs 1024"; do
  JAXSIM_AMD_SPECIALIZE=1 timeout 300 python tools/bench_c5.py $a 2>&1 | tail -1 | sed "s/^/$a: /" | tee -a "$OUT/c5.txt"
done
export JAXSIM_AMD_SPEC_EXTRA_FLAGS=-DJXS_PHASE_TIMING
for a in "4 4096" "4 4096 rigid standing"; do
  JAXSIM_AMD_SPECIALIZE=1 timeout 300 python tools/phase_timing_rigid.py $a 2>&1 | tee -a "$OUT/phases_c5.txt"
done
