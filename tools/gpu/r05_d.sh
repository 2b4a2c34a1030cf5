#!/usr/bin/env bash
# round 5, call D: the full GPU suite (both kernel policies, specialised objects required), smoke, the bench line and the
# contact-model figures after the move of the contact solves into the tree.
set -u
OUT=gpurun_out/r05_d
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x -n 4 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err; tail -c 400 $OUT/bench_steps20.json; echo
for a in "--points 4" "--points 16" "--contact relaxed --points 16" "--contact relaxed --points 32 --envs 1024" "--contact relaxed --points 32 --envs 1024 --standing" "--contact rigid --points 32 --envs 1024 --dtype float64" "--contact relaxed --points 32 --envs 1024 --dtype float64"; do
  echo "bench_c5 $a" >> $OUT/c5.txt
  JAXSIM_AMD_SPECIALIZE=1 timeout 600 python tools/bench_c5.py $a >> $OUT/c5.txt 2>> $OUT/c5.err
done
cat $OUT/c5.txt | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print('   ', round(d['ms_per_step'] * 1e3, 2), 'us', round(d['env_steps_per_s'] / 1e6, 2), 'M/s', d['workload'])
"
