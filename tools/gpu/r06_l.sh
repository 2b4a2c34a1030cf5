#!/usr/bin/env bash
# round 6, call L: the rigid contact modes with KParams::max_children bounding the child gathers (working tree), library
# and model-specialised kernels; headline sweep
set -u
OUT=$PWD/gpurun_out/r06_l
mkdir -p $OUT
one() {
  local pol=$1; shift
  JAXSIM_AMD_SPECIALIZE=$pol python tools/bench_c5.py "$@" 2> $OUT/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f us' % (d['ms_per_step']*1e3))"
}
for pol in 1 1; do
  echo "  policy $pol: c5 rigid 4pt: $(one $pol)   relaxed humanoid 32pt: $(one $pol --contact relaxed --points 32 --envs 1024)   relaxed quadruped 16pt: $(one $pol --contact relaxed --points 16)   c5 standing: $(one $pol --standing)" | tee -a $OUT/ab.log
done
JAXSIM_AMD_SPECIALIZE=1 python tools/sweep.py --sizes 1024,8192,65536 --steps 300 > $OUT/sweep.log 2>&1; cut -c40-120 $OUT/sweep.log
