#!/usr/bin/env bash
# round 5, call B: RelaxedRigidContacts solved IN THE TREE (jxs_rigid.inc ta_*) against round 4's link space and the dense
# triangles, alternating on one box.  Kernels pre-built by tools/gpu/r05_b_prebuild.py.
set -u
OUT=gpurun_out/r05_b
mkdir -p $OUT
export JAXSIM_AMD_SPECIALIZE=1
run() {  # name, env assignments..., -- bench_c5 args
  local name=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 600 python tools/bench_c5.py "$@" 2>> $OUT/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step']*1e3,2), 'us', round(d['env_steps_per_s']/1e6,2), 'M/s finite', d['finite_envs'])" | tee -a $OUT/ab.txt
}
for rep in 1 2; do
  run "humanoid32 tree     " X=1 -- --contact relaxed --points 32 --envs 1024
  run "humanoid32 linkspace" JXS_PREFER_LINKSPACE=1 -- --contact relaxed --points 32 --envs 1024
  run "humanoid32 stand tree" X=1 -- --contact relaxed --points 32 --envs 1024 --standing
  run "humanoid32 stand ls  " JXS_PREFER_LINKSPACE=1 -- --contact relaxed --points 32 --envs 1024 --standing
  run "quad16 tree         " X=1 -- --contact relaxed --points 16
  run "quad16 dense        " JXS_DISABLE_CT_TREE=1 -- --contact relaxed --points 16
  run "quad4 tree          " X=1 -- --contact relaxed --points 4
  run "quad4 dense         " JXS_DISABLE_CT_TREE=1 -- --contact relaxed --points 4
done
run "humanoid32 dense    " JXS_DISABLE_CT_TREE=1 JXS_DISABLE_LINKSPACE=1 -- --contact relaxed --points 32 --envs 1024
