#!/usr/bin/env bash
# round 6, call K: library kernels (JAXSIM_AMD_SPECIALIZE=0) against model-specialised ones (=1: built on the box where the
# tree has none) for the rigid contact modes, three trees, one box
set -u
OUT=$PWD/gpurun_out/r06_k
mkdir -p $OUT
one() {  # policy, args...
  local pol=$1; shift
  JAXSIM_AMD_SPECIALIZE=$pol python tools/bench_c5.py "$@" 2> $OUT/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f us' % (d['ms_per_step']*1e3))"
  grep -i "warn\|error\|generic" $OUT/err.txt | head -2
}
for d in tools/ab/wt_r5 tools/ab/wt_c2c29ac .; do
  ( cd $d
    echo "== $d"
    for pol in 0 1; do
      echo "  policy $pol: c5 rigid 4pt: $(one $pol)   relaxed humanoid 32pt: $(one $pol --contact relaxed --points 32 --envs 1024)   relaxed quadruped 16pt: $(one $pol --contact relaxed --points 16)"
    done
  ) 2>&1 | tee -a $OUT/ab.log
done
