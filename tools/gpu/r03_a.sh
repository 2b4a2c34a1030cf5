#!/usr/bin/env bash
# round 3, GPU call A: the two-wave workgroup variant -- parity suite, A/B timing, cycle stamps
set -u
R=$PWD
OUT=$R/gpurun_out/r03_a
mkdir -p "$OUT"
timeout 900 python -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/summary.txt"
tail -3 "$OUT/pytest.log" | tee -a "$OUT/summary.txt"
for duo in 0 1; do
  JXS_DUO=$duo timeout 300 python tools/sweep.py --sizes 1024,2048,4096 --steps 1000 2>&1 | sed "s/^/JXS_DUO=$duo /" | tee -a "$OUT/summary.txt"
done
JXS_DUO=0 JAXSIM_AMD_SPECIALIZE=0 timeout 300 python tools/sweep.py --sizes 1024 --steps 1000 2>&1 | sed "s/^/generic JXS_DUO=0 /" | tee -a "$OUT/summary.txt"
JXS_DUO=1 JAXSIM_AMD_SPECIALIZE=0 timeout 300 python tools/sweep.py --sizes 1024 --steps 1000 2>&1 | sed "s/^/generic JXS_DUO=1 /" | tee -a "$OUT/summary.txt"
export JAXSIM_AMD_SPEC_EXTRA_FLAGS=-DJXS_PHASE_TIMING
JXS_DUO=1 JAXSIM_AMD_SPECIALIZE=1 timeout 300 python tools/phase_timing.py > "$OUT/phases_duo.log" 2>&1
JXS_DUO=0 JAXSIM_AMD_SPECIALIZE=1 timeout 300 python tools/phase_timing.py > "$OUT/phases_solo.log" 2>&1
unset JAXSIM_AMD_SPEC_EXTRA_FLAGS
cat "$OUT/phases_duo.log" | tee -a "$OUT/summary.txt"
timeout 600 python bench.py --steps 20 --warmup 5 > "$OUT/bench_steps20.json" 2> "$OUT/bench_steps20.err"
tail -c 1500 "$OUT/bench_steps20.json"
