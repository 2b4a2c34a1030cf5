#!/usr/bin/env bash
# round 6, call B: system_dynamics / link_contact_forces and the height-field terrain through the library's kernels
set -u
OUT=gpurun_out/r06_b
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "(system_dynamics or link_contact or link_forces_from or height_field or plane_terrain) and not specialised" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/pytest.log
