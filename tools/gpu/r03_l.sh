#!/usr/bin/env bash
# round 3, GPU call L: MFMA trailing update of the contact solvers' blocked Cholesky
set -u
R=$PWD
OUT=$R/gpurun_out/r03_l
mkdir -p "$OUT"
timeout 900 python -m pytest tests -m gpu -q > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"
tail -8 "$OUT/pytest.log"
for m in 0 1; do
for a in "--points 16" "--points 16 --standing" "--contact relaxed --points 16" "--contact relaxed --points 16 --standing" "--contact relaxed --points 32 --envs 1024" "--contact relaxed --points 32 --envs 1024 --standing"; do
  JXS_NO_MFMA=$m JAXSIM_AMD_SPECIALIZE=1 timeout 300 python tools/bench_c5.py $a 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('JXS_NO_MFMA=$m', '$a', '%.1f us' % (d['ms_per_step']*1e3), 'finite', d['finite_envs'])" | tee -a "$OUT/c5.txt"
done
done
