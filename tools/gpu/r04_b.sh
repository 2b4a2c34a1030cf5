#!/usr/bin/env bash
# round 4, call B: ubench (with T11), the whole GPU suite in both kernel policies, bench.py default + --steps 20
set -u
R=$PWD; OUT=$R/gpurun_out/r04_b; mkdir -p "$OUT"
timeout 120 tools/ubench/exec_dpp 1024 > "$OUT/exec_dpp.txt" 2>&1; echo "ubench rc=$?"
timeout 900 python -m pytest tests -m gpu -q -x > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest.log"
timeout 600 python bench.py > "$OUT/bench_N1.json" 2> "$OUT/bench_N1.err"; echo "bench rc=$?"
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_steps20.json" 2> "$OUT/bench_steps20.err"; echo "bench20 rc=$?"
python - <<'P'
import json
for f in ("bench_N1","bench_steps20"):
    try:
        d=json.load(open(f"gpurun_out/r04_b/{f}.json"))
        print(f, d["value"], d["ms_per_step"], d.get("roofline",{}).get("frac"))
        for k,v in d.get("other_contact_models",{}).items() if isinstance(d.get("other_contact_models"),dict) else []:
            print("  ",k, {kk:vv for kk,vv in v.items() if kk in ("ms_per_step","value","finite_envs")} if isinstance(v,dict) else v)
    except Exception as e: print(f,"ERR",e)
P
