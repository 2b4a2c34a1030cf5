#!/usr/bin/env bash
# round 6, call I: the driver's end-of-round commands on the round's sources -- the whole GPU suite (library and
# model-specialised kernels: every object pre-built), smoke(), the bench line (default, --steps 20, batch 8192)
set -u
OUT=gpurun_out/r06_i
mkdir -p $OUT
JXS_ERR_LOG=$PWD/$OUT/errors.log timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=6 > $OUT/pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$?"; tail -12 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $OUT/smoke.log
python bench.py > $OUT/bench_N1.json 2> $OUT/bench_N1.err; echo "bench rc=$?"
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err; echo "bench steps20 rc=$?"
python - <<'PY'
import json
for f in ('bench_N1','bench_steps20'):
    d=json.loads(open(f'gpurun_out/r06_i/{f}.json').read().strip().splitlines()[-1])
    r=d['roofline']
    g=lambda k,kk: (d.get(k) or {}).get(kk)
    print(f, round(d['ms_per_step']*1e3,3),'us', round(d['value']/1e6,1),'M; steady', g('steady_state','us_per_step'), 'rollout', g('fused_rollout','us_per_step'), 'roofline', round(r['frac'],4), r['kernel_avg_launch_us'], 'traffic', r['traffic'])
    for k in ('global_batch_8192_one_gpu','saturated','saturated_4x'):
        print('  ',k, g(k,'us_per_step'), g(k,'env_steps_per_s'), g(k,'valu_issue_util'), g(k,'error'))
    print('   cpu', (d.get('cpu_baseline') or {}).get('value'))
    print('   other', {k:(v.get('us_per_step') if isinstance(v,dict) else None) for k,v in (d.get('other_contact_models') or {}).items()})
PY
