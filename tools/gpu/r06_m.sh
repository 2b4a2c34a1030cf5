#!/usr/bin/env bash
# round 6, call M: the query-kernel table incl. the reference's three system_dynamics contact benchmarks, and the two bench
# lines with the PMC traffic of profiles/r06_pmc.json (same kernel sources)
set -u
OUT=gpurun_out/r06_m
mkdir -p $OUT
JAXSIM_AMD_SPECIALIZE=1 python tools/bench_queries.py --specialised > $OUT/query_kernels.txt 2> $OUT/query_kernels.err; echo "queries rc=$?"; cat $OUT/query_kernels.txt; tail -3 $OUT/query_kernels.err
python bench.py > $OUT/bench_N1.json 2> $OUT/bench_N1.err; echo "bench rc=$?"
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err; echo "bench steps20 rc=$?"
python - <<'PY'
import json
for f in ('bench_N1','bench_steps20'):
    d=json.loads(open(f'gpurun_out/r06_m/{f}.json').read().strip().splitlines()[-1])
    r=d['roofline']
    print(f, round(d['ms_per_step']*1e3,3),'us', round(d['value']/1e6,1),'M; roofline', round(r['frac'],4), r['kernel_avg_launch_us'], 'traffic', r['traffic'], r.get('traffic_source'))
PY
