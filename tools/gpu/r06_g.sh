#!/usr/bin/env bash
# round 6, call G: links with up to twelve children (octopod, twelve-spoke hub), the tightened random-tree gate of the
# RigidContacts tree solve, and the headline sweep on the new sources (kMaxChildren 6 -> 12 must not move it)
set -u
OUT=gpurun_out/r06_g
mkdir -p $OUT
JXS_ERR_LOG=$PWD/$OUT/errors.log timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "(more_than_six or octopod or contact_tree_solve_on_random_trees) and library" > $OUT/pytest_hub.log 2>&1; echo "pytest hub rc=$?"; tail -15 $OUT/pytest_hub.log
cat $OUT/errors.log 2>/dev/null | tail -30
JAXSIM_AMD_SPECIALIZE=1 python tools/sweep.py --sizes 1024,8192,65536 --steps 300 > $OUT/sweep.log 2>&1; cut -c40-120 $OUT/sweep.log
