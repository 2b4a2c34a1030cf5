#!/usr/bin/env bash
# round 5, call E: GPU suite (no -x), the bench line with `python_step_loop`
set -u
OUT=gpurun_out/r05_e
mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q -n 4 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest.log
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --saturated-envs 0 --no-other-contact-models > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err
python -c "
import json; d=json.load(open('$OUT/bench_steps20.json')); print('value', d['value']/1e6, 'M', d['ms_per_step']*1e3, 'us'); print('python_step_loop', json.dumps(d.get('python_step_loop'))); print('steady', d['steady_state'])"
