#!/usr/bin/env bash
# round 4, call N: fp64 sincos -- the suite through the library kernels (record mode), fp64 timings
set -u
T=1100 tools/gpu/r04_record_manifest.sh 2>&1 | tail -3
export JAXSIM_AMD_SPECIALIZE=1
python bench.py --gpus 1 --dtype float64 --no-cpu-baseline --saturated-envs 0 --no-other-contact-models 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp64 headline step', d['ms_per_step']*1e3, 'us  steady', (d.get('steady_state') or {}).get('us_per_step'))"
timeout 400 python tools/bench_c5.py --contact relaxed --points 32 --envs 1024 --dtype float64 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp64 relaxed humanoid', round(d.get('ms_per_step')*1e3,1), 'us')"
