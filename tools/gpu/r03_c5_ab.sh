#!/usr/bin/env bash
# round 3: A/B of a packer knob on config 5 (tools/bench_c5.py) and the 16-point / relaxed variants, one box
set -u
R=$PWD
OUT=$R/gpurun_out/${1:-r03_c5_ab}
mkdir -p "$OUT"
if [ "${PYTEST:-0}" = 1 ]; then
  timeout 900 python -m pytest tests -m gpu -q -x > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"
  tail -3 "$OUT/pytest.log"
fi
for i in 1 2 3; do
  for v in "" "$2"; do
    env $v JAXSIM_AMD_SPECIALIZE=1 timeout 300 python tools/bench_c5.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('${v:-default}', 'c5 %.2f us' % (d['ms_per_step']*1e3))" | tee -a "$OUT/summary.txt"
  done
done
for v in "" "$2"; do
  env $v JAXSIM_AMD_SPECIALIZE=1 timeout 300 python tools/bench_c5.py --points 16 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('${v:-default}', '16 points %.2f us' % (d['ms_per_step']*1e3))" | tee -a "$OUT/summary.txt"
  env $v JAXSIM_AMD_SPECIALIZE=1 timeout 300 python tools/bench_c5.py --contact relaxed --points 16 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('${v:-default}', 'relaxed 16 points %.2f us' % (d['ms_per_step']*1e3))" | tee -a "$OUT/summary.txt"
done
