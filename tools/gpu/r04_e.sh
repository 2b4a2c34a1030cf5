#!/usr/bin/env bash
# round 4, call E: suite (require policy) + bench + relaxed / config-5 timings
set -u
R=$PWD; OUT=$R/gpurun_out/r04_e; mkdir -p "$OUT"
timeout 900 python -m pytest tests -m gpu -q > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -6 "$OUT/pytest.log"
timeout 600 python bench.py > "$OUT/bench_N1.json" 2> "$OUT/bench_N1.err"; echo "bench rc=$?"
python - <<'P'
import json
d=json.load(open("gpurun_out/r04_e/bench_N1.json"))
print("headline", d["value"], d["ms_per_step"], {k:d["roofline"].get(k) for k in ("frac","valu_per_wave_static","valu_issue_util","lane_slot_efficiency")})
print("saturated", {k:d["saturated"].get(k) for k in ("us_per_step","env_steps_per_s","finite_envs","valu_issue_util")})
for k,v in d["other_contact_models"].items():
    if isinstance(v,dict): print(k, v.get("us_per_step"), v.get("env_steps_per_s"), v.get("finite_envs"))
P
for dis in 0 1; do
  if [ $dis = 1 ]; then export JXS_DISABLE_LINKSPACE=1; fi
  JAXSIM_AMD_SPECIALIZE=1 timeout 400 python tools/bench_c5.py --contact relaxed --points 32 --envs 1024 --standing 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('relaxed humanoid? disable_linkspace=$dis', d.get('ms_per_step'), d.get('finite_envs'))"
done
