#!/usr/bin/env bash
# round 5, call Z: what the driver runs at the end of the round, on the final tree
set -u
OUT=gpurun_out/r05_z
mkdir -p $OUT
timeout 1500 python -m pytest tests/ -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
( time python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; grep real $OUT/bench_default.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err; echo "bench20 rc=$?"
python -c "
import json
for f in ('bench_default','bench_steps20'):
    d=json.loads(open('$OUT/'+f+'.json').read().strip().splitlines()[-1])
    print(f, round(d['value']/1e6,2),'M', round(d['ms_per_step']*1e3,3),'us', 'roofline', round(d['roofline']['frac'],4), 'traffic', d['roofline']['traffic'], 'cpu', round(d['cpu_baseline']['value']/1e6,2), 'M', d['cpu_baseline']['cores'])
"
