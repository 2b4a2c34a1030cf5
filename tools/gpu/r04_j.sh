#!/usr/bin/env bash
# round 4, call J: how the timed region learns of its end -- a polled word written by the stream against hipStreamQuery
set -u
export JAXSIM_AMD_SPECIALIZE=1
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --saturated-envs 0 --no-other-contact-models > /dev/null 2>&1  # builds the kernel
for rep in 1 2 3; do
  for q in 0 1; do
    if [ $q = 1 ]; then export JXS_TIMED_WAIT_QUERY=1; else unset JXS_TIMED_WAIT_QUERY; fi
    python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --saturated-envs 0 --no-other-contact-models 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('query=$q', round(d['value']/1e6,2), 'M', d['ms_per_step']*1e3, 'us; steady', d['steady_state'].get('us_per_step') if isinstance(d.get('steady_state'),dict) else None)"
  done
done
