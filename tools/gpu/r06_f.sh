#!/usr/bin/env bash
# round 6, call F: 64-word records (12.4 KB per humanoid wave = ten LDS granules, twelve waves per CU): residency, batch sweep,
# the GPU suite under the library policy
set -u
OUT=gpurun_out/r06_f
mkdir -p $OUT
JAXSIM_AMD_SPECIALIZE=1 python tools/sweep.py --sizes 1024,2048,4096,6144,8192,12288,16384,65536,262144 --steps 300 > $OUT/sweep.log 2>&1; cut -c40-120 $OUT/sweep.log
for n in 1024 8192 65536; do
  JAXSIM_AMD_SPEC_EXTRA_FLAGS=-DJXS_PHASE_TIMING JAXSIM_AMD_SPECIALIZE=1 python tools/phase_timing.py $n > $OUT/phases_$n.log 2>&1
  echo "== N=$n"; grep -E "total|placement|residency" $OUT/phases_$n.log
done
timeout 1500 python -m pytest tests/ -q -m gpu -k "not specialised" -x --deselect tests/test_specialize.py --deselect tests/test_bench_gpu.py > $OUT/pytest_lib.log 2>&1; echo "pytest lib rc=$?"; tail -3 $OUT/pytest_lib.log
