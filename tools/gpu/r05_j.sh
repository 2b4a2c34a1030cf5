#!/usr/bin/env bash
# round 5, call J: the GPU suite with the specialised objects required, smoke, and the round's rocprofv3 evidence
set -u
OUT=gpurun_out/r05_j
mkdir -p $OUT
rm -f $OUT/err_log.txt
JXS_ERR_LOG=$PWD/$OUT/err_log.txt timeout 1200 python -m pytest tests -m gpu -q -n 4 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
bash tools/profile_round.sh r05_prof r05 > $OUT/profile_round.log 2>&1; tail -5 $OUT/profile_round.log
