#!/usr/bin/env bash
# round 3, GPU calls T/U: step kernel variants A/B on one box (compiler scheduling strategies, developer knobs); labels and
# flags come from the arguments:  r03_t.sh OUTDIR "label=flags" ...
set -u
R=$PWD
OUT=$R/gpurun_out/${1:-r03_t}
shift
mkdir -p "$OUT"
run() {  # label, extra flags
  JAXSIM_AMD_SPEC_EXTRA_FLAGS="$2" JAXSIM_AMD_SPECIALIZE=1 timeout 300 python tools/sweep.py --sizes 1024,2048 --steps 1000 2>&1 | sed "s/^/$1: /" | tee -a "$OUT/summary.txt"
}
if [ "${PYTEST:-0}" = 1 ]; then
  timeout 900 python -m pytest tests -m gpu -q -x > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"
  tail -3 "$OUT/pytest.log"
fi
for i in 1 2; do
  run "default" ""
  for spec in "$@"; do run "${spec%%=*}" "${spec#*=}"; done
done
