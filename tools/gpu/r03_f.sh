#!/usr/bin/env bash
# round 3, GPU call F: full parity suite after the table / policy / bench changes; config 5 at 16 vs 32 lanes per environment
set -u
R=$PWD
OUT=$R/gpurun_out/r03_f
mkdir -p "$OUT"
timeout 900 python -m pytest tests -m gpu -q > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"
tail -6 "$OUT/pytest.log"
for g in 16 32; do
  for a in "" "--standing"; do
    JXS_MIN_LANES=$g JAXSIM_AMD_SPECIALIZE=1 timeout 300 python tools/bench_c5.py $a 2>&1 | tail -1 | sed "s/^/lanes>=$g $a: /" | tee -a "$OUT/c5.txt"
  done
  JXS_MIN_LANES=$g JAXSIM_AMD_SPECIALIZE=1 timeout 300 python tools/bench_c5.py --envs 16384 2>&1 | tail -1 | sed "s/^/lanes>=$g N=16384: /" | tee -a "$OUT/c5.txt"
  JXS_MIN_LANES=$g JAXSIM_AMD_SPECIALIZE=1 timeout 300 python tools/bench_c5.py --points 16 2>&1 | tail -1 | sed "s/^/lanes>=$g 16 points: /" | tee -a "$OUT/c5.txt"
done
timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/bench_steps20.json" 2> "$OUT/bench_steps20.err"; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03_f/bench_steps20.json"))
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["config"].get("specialised_object"), d.get("default_contact_params"), d["steady_state"]["us_per_step"])
print({k:(v.get("us_per_step"), v.get("roofline",{}).get("frac")) for k,v in d["other_contact_models"].items() if isinstance(v,dict)})
PY
tail -3 "$OUT/bench_steps20.err"
