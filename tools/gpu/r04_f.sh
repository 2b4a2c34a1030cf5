#!/usr/bin/env bash
# round 4, call F: RigidContacts in fp64, humanoid with 32 points: link space against the LDS triangles
set -u
for dis in 0 1; do
  if [ $dis = 1 ]; then export JXS_DISABLE_LINKSPACE=1; fi
  for st in "" "--standing"; do
    JAXSIM_AMD_SPECIALIZE=cached timeout 400 python tools/bench_c5.py --contact rigid --points 32 --envs 1024 --dtype float64 --steps 100 $st 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rigid fp64 humanoid 32 points disable_linkspace=$dis $st', round(d['ms_per_step']*1e3,1), 'us', d['finite_envs'])"
  done
done
