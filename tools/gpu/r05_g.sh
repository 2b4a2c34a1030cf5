#!/usr/bin/env bash
# round 5, call G: full GPU suite (both policies) with the measured errors logged, smoke, bench lines, fp32 error table, contact-model figures
set -u
OUT=gpurun_out/r05_g
mkdir -p $OUT
rm -f $OUT/err_log.txt
JXS_ERR_LOG=$PWD/$OUT/err_log.txt timeout 1200 python -m pytest tests -m gpu -q -n 4 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err
python bench.py > $OUT/bench_N1.json 2> $OUT/bench_N1.err
python -c "
import json
for f in ('bench_steps20','bench_N1'):
    d=json.load(open('$OUT/'+f+'.json')); print(f, 'value', round(d['value']/1e6,2), 'M', round(d['ms_per_step']*1e3,3), 'us', 'roofline', d['roofline']['frac'], 'py loop', d.get('python_step_loop'))
    print('   other', json.dumps(d.get('other_contact_models'))[:1500])
"
timeout 900 python tools/fp32_error_gpu.py 512 > $OUT/fp32_error_gpu.txt 2>&1; tail -5 $OUT/fp32_error_gpu.txt
