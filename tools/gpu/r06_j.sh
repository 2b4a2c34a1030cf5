#!/usr/bin/env bash
# round 6, call J: where did the rigid contact modes lose 8 - 17 % between round 5 and now?  The same benchmarks from three
# trees on one box: round 5's final commit, the commit before this session (c2c29ac), the working tree.
set -u
OUT=$PWD/gpurun_out/r06_j
mkdir -p $OUT
for d in tools/ab/wt_r5 tools/ab/wt_c2c29ac .; do
  ( cd $d
    echo "== $d"
    for rep in 1 2; do
      JAXSIM_AMD_SPECIALIZE=1 python tools/bench_c5.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  c5 rigid 4pt N=4096: %.2f us' % (d['ms_per_step']*1e3))"
      JAXSIM_AMD_SPECIALIZE=1 python tools/bench_c5.py --contact relaxed --points 32 --envs 1024 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  relaxed humanoid 32pt N=1024: %.2f us' % (d['ms_per_step']*1e3))"
      JAXSIM_AMD_SPECIALIZE=1 python tools/bench_c5.py --contact relaxed --points 16 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  relaxed quadruped 16pt N=4096: %.2f us' % (d['ms_per_step']*1e3))"
    done
  ) 2>&1 | tee -a $OUT/ab.log
done
