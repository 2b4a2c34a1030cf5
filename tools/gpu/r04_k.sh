#!/usr/bin/env bash
# round 4, call K: RelaxedRigidContacts humanoid (32 points) in fp64 -- the reference's default precision -- with the
# reference's default parameters (mu = 0.005) and with the estimated ones: link space against the dense path
set -u
for p in "--default-params" ""; do
  for dis in 0 1; do
    if [ $dis = 1 ]; then export JXS_DISABLE_LINKSPACE=1; else unset JXS_DISABLE_LINKSPACE; fi
    for st in "" "--standing"; do
      JAXSIM_AMD_SPECIALIZE=1 timeout 400 python tools/bench_c5.py --contact relaxed --points 32 --envs 1024 --dtype float64 $p $st 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp64 relaxed humanoid [$p] [$st] disable_linkspace=$dis', round(d.get('ms_per_step')*1e3,1), 'us', d.get('finite_envs'))"
    done
  done
done
