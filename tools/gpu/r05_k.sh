#!/usr/bin/env bash
# round 5, call K: the bench line after the second pass over the host path of step(); RigidContacts fp64 humanoid on the
# triangles (the default again); the 200-point quadruped in 32- and 64-lane groups
set -u
OUT=gpurun_out/r05_k
mkdir -p $OUT
export JAXSIM_AMD_SPECIALIZE=1
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --saturated-envs 0 --no-other-contact-models > $OUT/bench_steps20.json 2> $OUT/bench.err
python -c "import json; d=json.load(open('$OUT/bench_steps20.json')); print('value', round(d['value']/1e6,2), d['python_step_loop'])"
timeout 600 python tools/bench_c5.py --contact rigid --points 32 --envs 1024 --dtype float64 >> $OUT/c5.txt 2>> $OUT/err.log
for rep in 1 2; do
  for g in 0 64; do
    if [ $g = 64 ]; then export JXS_MIN_LANES=64; else unset JXS_MIN_LANES; fi
    timeout 600 python tools/bench_c5.py --contact relaxed --points 200 >> $OUT/c5.txt 2>> $OUT/err.log
  done
done
unset JXS_MIN_LANES
python -c "
import json
for ln in open('$OUT/c5.txt'):
    d=json.loads(ln); print(round(d['ms_per_step']*1e3,1),'us', round(d['env_steps_per_s']/1e6,2),'M/s lanes',d['lanes_per_env'], d['workload'][:70])"
