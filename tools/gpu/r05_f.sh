#!/usr/bin/env bash
# round 5, call F: config 5 (quadruped, RigidContacts, 4 points, N = 4096) with the interior-point solver skipping the
# blocks of points that are active in no environment of the wave, against the full twelve columns; random states (the
# bench's) and standing states (all four feet down: nothing to skip, the guards only cost).
set -u
OUT=gpurun_out/r05_f
mkdir -p $OUT
export JAXSIM_AMD_SPECIALIZE=1
for rep in 1 2 3; do
  for f in "" "-DJXS_NO_QP_BLOCK_SKIP"; do
    for st in "" "--standing"; do
      JAXSIM_AMD_SPEC_EXTRA_FLAGS="$f" timeout 300 python tools/bench_c5.py --points 4 $st 2>> $OUT/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('flags=[$f] $st', round(d['ms_per_step']*1e3,2), 'us', round(d['env_steps_per_s']/1e6,2), 'M/s')" | tee -a $OUT/ab.txt
    done
  done
done
