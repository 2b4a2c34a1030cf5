#!/usr/bin/env bash
# round 3, GPU call S: base solve of the row-distributed sweeps as a Gauss-Jordan over the row lanes (A/B on one box)
set -u
R=$PWD
OUT=$R/gpurun_out/r03_s
mkdir -p "$OUT"
timeout 900 python -m pytest tests -m gpu -q -x > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"
tail -3 "$OUT/pytest.log"
for i in 1 2; do
JXS_DUO=0 timeout 300 python tools/sweep.py --sizes 1024,2048 --steps 1000 2>&1 | tee -a "$OUT/summary.txt"
JAXSIM_AMD_SPEC_EXTRA_FLAGS=${AB_FLAG:--DJXS_NO_ROW_BASE_SOLVE} JAXSIM_AMD_SPECIALIZE=1 timeout 300 python tools/sweep.py --sizes 1024,2048 --steps 1000 2>&1 | sed "s/^/B: /" | tee -a "$OUT/summary.txt"
done
JXS_DUO=1 timeout 300 python tools/sweep.py --sizes 1024 --steps 1000 2>&1 | sed "s/^/JXS_DUO=1 /" | tee -a "$OUT/summary.txt"
