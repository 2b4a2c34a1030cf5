#!/usr/bin/env bash
# round 6, call H: record every kernel description the GPU suite asks for on the round's final kernel sources
# (tests/spec_manifest.txt <- gpurun_out/r06_h/spec_manifest.txt; __graft_entry__.build() pre-builds them), with the
# measured errors of the suite's noted cases
set -u
OUT=gpurun_out/r06_h
mkdir -p $OUT
rm -f $OUT/spec_manifest.raw
JXS_ERR_LOG=$PWD/$OUT/errors.log JAXSIM_AMD_TEST_RECORD=1 JAXSIM_AMD_SPEC_RECORD=$PWD/$OUT/spec_manifest.raw timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $OUT/record_pytest.log 2>&1
echo "pytest rc=$?"; tail -25 $OUT/record_pytest.log
sort -u $OUT/spec_manifest.raw > $OUT/spec_manifest.txt; wc -l $OUT/spec_manifest.txt; rm -f $OUT/spec_manifest.raw
