#!/usr/bin/env bash
# round 3, GPU call M: default build after making the matrix-core Cholesky an opt-in build; full parity suite
set -u
R=$PWD
OUT=$R/gpurun_out/r03_o
mkdir -p "$OUT"
timeout 900 python -m pytest tests -m gpu -q > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"
tail -4 "$OUT/pytest.log"
for a in "" "--standing" "--points 16" "--points 16 --standing" "--contact relaxed" "--contact relaxed --points 16" "--contact relaxed --points 32 --envs 1024" "--contact relaxed --points 32 --envs 1024 --standing"; do
  JAXSIM_AMD_SPECIALIZE=1 timeout 300 python tools/bench_c5.py $a 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$a', '%.1f us' % (d['ms_per_step']*1e3), 'finite', d['finite_envs'])" | tee -a "$OUT/c5.txt"
done
