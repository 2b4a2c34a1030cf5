#!/usr/bin/env bash
# round 6, call C: one record per LINK in the LDS (13.2 KB per humanoid wave, three waves per SIMD): the whole GPU suite under
# the library policy, then the bench line with the specialised headline kernel (batch 1024, 8192 and 65536 on this GPU)
set -u
OUT=gpurun_out/r06_c
mkdir -p $OUT
timeout 1500 python -m pytest tests/ -q -m gpu -k "not specialised" -x --deselect tests/test_specialize.py --deselect tests/test_bench_gpu.py > $OUT/pytest_lib.log 2>&1; echo "pytest lib rc=$?"; tail -5 $OUT/pytest_lib.log
python bench.py --gpus 1 --no-cpu-baseline --no-other-contact-models --no-python-loop > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python bench.py --gpus 1 --global-batch 8192 --no-cpu-baseline --no-other-contact-models --no-python-loop --saturated-envs 0 > $OUT/bench8192.json 2> $OUT/bench8192.err; echo "bench8192 rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_c/bench.json').read().strip().splitlines()[-1])
print('1024:', round(d['ms_per_step']*1e3,3),'us', round(d['value']/1e6,1),'M; steady', d['steady_state']['us_per_step'], 'rollout', d['fused_rollout']['us_per_step'])
for k in ('global_batch_8192_one_gpu','saturated'):
    s=d.get(k) or {}
    print(k, s.get('us_per_step'), s.get('env_steps_per_s'), s.get('valu_issue_util'), s.get('error'))
d=json.loads(open('gpurun_out/r06_c/bench8192.json').read().strip().splitlines()[-1])
print('8192 line:', round(d['ms_per_step']*1e3,3),'us', round(d['value']/1e6,1),'M; steady', d['steady_state']['us_per_step'], 'rollout', d['fused_rollout']['us_per_step'])
PY
