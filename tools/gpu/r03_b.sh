#!/usr/bin/env bash
# round 3, GPU call B: which per-CU resources the waves of the step kernel share (placement + stream costs), the
# two-wave variant's stamps with placement, the parity suite with per-model gates and the measured-error log
set -u
R=$PWD
OUT=$R/gpurun_out/r03_b
mkdir -p "$OUT"
timeout 120 tools/ubench/cu_share > "$OUT/cu_share.txt" 2>&1; cat "$OUT/cu_share.txt"
timeout 60 tools/ubench/wg_simd > "$OUT/wg_simd.txt" 2>&1; cat "$OUT/wg_simd.txt"
export JAXSIM_AMD_SPEC_EXTRA_FLAGS=-DJXS_PHASE_TIMING
JXS_DUO=1 JAXSIM_AMD_SPECIALIZE=1 timeout 300 python tools/phase_timing.py > "$OUT/phases_duo.log" 2>&1
JXS_DUO=0 JAXSIM_AMD_SPECIALIZE=1 timeout 300 python tools/phase_timing.py > "$OUT/phases_solo.log" 2>&1
unset JAXSIM_AMD_SPEC_EXTRA_FLAGS
tail -8 "$OUT/phases_duo.log"; tail -4 "$OUT/phases_solo.log"
rm -f "$OUT/errors.log"
JXS_ERR_LOG="$OUT/errors.log" timeout 900 python -m pytest tests -m gpu -q > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"
tail -15 "$OUT/pytest.log"
JXS_DUO=0 JXS_ERR_LOG="$OUT/errors_solo.log" timeout 900 python -m pytest tests -m gpu -q -x > "$OUT/pytest_solo.log" 2>&1; echo "pytest (JXS_DUO=0) rc=$?"
tail -3 "$OUT/pytest_solo.log"
cat "$OUT/errors.log"
