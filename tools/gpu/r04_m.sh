#!/usr/bin/env bash
# round 4, call M: the recorded rollout -- record run of the suite, headline check, timing of recording
set -u
T=1100 tools/gpu/r04_record_manifest.sh 2>&1 | tail -3
export JAXSIM_AMD_SPECIALIZE=1
for i in 1 2 3; do
python bench.py --gpus 1 --no-cpu-baseline --saturated-envs 0 --no-other-contact-models 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline', d['ms_per_step']*1e3, 'steady', d['steady_state'].get('us_per_step'), 'fused', d['fused_rollout'])" | cut -c1-300
done
