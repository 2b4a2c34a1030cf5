#!/usr/bin/env bash
# round 4, call D: fp32 error table against the clean fp64 truth + the suite with error log
set -u
R=$PWD; OUT=$R/gpurun_out/r04_d; mkdir -p "$OUT"
timeout 600 python tools/fp32_error_gpu.py 512 > "$OUT/fp32_error_gpu.txt" 2>&1; echo "fp32 err rc=$?"
grep -v Warning "$OUT/fp32_error_gpu.txt" | grep seed
JXS_ERR_LOG="$OUT/errors.log" timeout 900 python -m pytest tests -m gpu -q > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -12 "$OUT/pytest.log"
