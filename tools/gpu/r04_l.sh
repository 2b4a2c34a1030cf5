#!/usr/bin/env bash
# round 4, call L: fp64 with hardware reciprocal / rsqrt seeds + Newton steps instead of IEEE division / sqrt
set -u
T=1100 tools/gpu/r04_record_manifest.sh 2>&1 | tail -3
export JAXSIM_AMD_SPECIALIZE=1
python bench.py --gpus 1 --no-cpu-baseline --saturated-envs 0 --no-other-contact-models 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp32', d['ms_per_step']*1e3, 'fp64', d['other_precision'])" | cut -c1-400
for st in "" "--standing"; do
  timeout 400 python tools/bench_c5.py --contact relaxed --points 32 --envs 1024 --dtype float64 $st 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp64 relaxed humanoid [$st]', round(d.get('ms_per_step')*1e3,1), 'us', d.get('finite_envs'))"
done
timeout 400 python tools/bench_c5.py --contact rigid --points 32 --envs 1024 --dtype float64 --standing 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp64 rigid humanoid standing', round(d.get('ms_per_step')*1e3,1), 'us', d.get('finite_envs'))"
