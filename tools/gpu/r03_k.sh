#!/usr/bin/env bash
# round 3, GPU call K: packed small solver, 96-bit LDS reads in the blocked Cholesky
set -u
R=$PWD
OUT=$R/gpurun_out/r03_k
mkdir -p "$OUT"
timeout 900 python -m pytest tests -m gpu -q > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"
tail -5 "$OUT/pytest.log"
for a in "" "--standing" "--points 16" "--points 16 --standing" "--contact relaxed --points 16" "--contact relaxed --points 32 --envs 1024" "--contact relaxed --points 32 --envs 1024 --standing"; do
  JAXSIM_AMD_SPECIALIZE=1 timeout 300 python tools/bench_c5.py $a 2>&1 | tail -1 | sed "s/^/$a: /" | tee -a "$OUT/c5.txt"
done
