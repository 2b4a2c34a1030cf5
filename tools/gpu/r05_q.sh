#!/usr/bin/env bash
# round 5, call Q: final validation on the committed sources -- GPU suite (specialised objects required), smoke, the
# round's rocprofv3 evidence, and the saturated config-5 figure
set -u
OUT=gpurun_out/r05_q
mkdir -p $OUT
rm -f $OUT/err_log.txt
JXS_ERR_LOG=$PWD/$OUT/err_log.txt timeout 1200 python -m pytest tests -m gpu -q -n 4 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
bash tools/profile_round.sh r05_prof r05 > $OUT/profile_round.log 2>&1
export JAXSIM_AMD_SPECIALIZE=1
for a in "--points 4 --envs 65536 --steps 60" "--points 4 --envs 16384 --steps 100" "--points 4 --standing" "--contact relaxed --points 200" "--contact relaxed --points 16" "--contact relaxed --points 32 --envs 1024"; do
  timeout 600 python tools/bench_c5.py $a >> $OUT/c5.txt 2>> $OUT/err.log
done
python -c "
import json
for ln in open('$OUT/c5.txt'):
    d=json.loads(ln); print(round(d['ms_per_step']*1e3,1),'us', round(d['env_steps_per_s']/1e6,2),'M/s N',d['envs'], d['workload'][:80])"
