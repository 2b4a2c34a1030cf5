#!/usr/bin/env bash
# round 4, call G: merged link-space sweeps (rl_merge) on/off for the relaxed humanoid + the link-space GPU tests, then the manifest record
set -u
R=$PWD; OUT=$R/gpurun_out/r04_g; mkdir -p "$OUT"
for dis in 0 1; do
  if [ $dis = 1 ]; then export JXS_DISABLE_RL_MERGE=1; fi
  JAXSIM_AMD_SPECIALIZE=1 timeout 400 python tools/bench_c5.py --contact relaxed --points 32 --envs 1024 --standing 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('relaxed humanoid disable_rl_merge=$dis', d.get('ms_per_step'), d.get('finite_envs'))"
  JAXSIM_AMD_SPECIALIZE=1 timeout 400 python tools/bench_c5.py --contact relaxed --points 32 --envs 1024 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('relaxed humanoid (falling) disable_rl_merge=$dis', d.get('ms_per_step'), d.get('finite_envs'))"
done
unset JXS_DISABLE_RL_MERGE
T=1100 tools/gpu/r04_record_manifest.sh
