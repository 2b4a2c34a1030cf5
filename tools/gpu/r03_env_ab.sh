#!/usr/bin/env bash
# round 3: A/B of a packer knob (environment variable) on one box:  r03_env_ab.sh OUTDIR VAR=VALUE
set -u
R=$PWD
OUT=$R/gpurun_out/${1:-r03_env_ab}
mkdir -p "$OUT"
if [ "${PYTEST:-0}" = 1 ]; then
  timeout 900 python -m pytest tests -m gpu -q -x > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"
  tail -3 "$OUT/pytest.log"
fi
for i in 1 2 3; do
  JAXSIM_AMD_SPECIALIZE=1 timeout 300 python tools/sweep.py --sizes 1024,2048 --steps 1000 2>&1 | sed "s/^/default: /" | tee -a "$OUT/summary.txt"
  env "$2" JAXSIM_AMD_SPECIALIZE=1 timeout 300 python tools/sweep.py --sizes 1024,2048 --steps 1000 2>&1 | sed "s/^/$2: /" | tee -a "$OUT/summary.txt"
done
