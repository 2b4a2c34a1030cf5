#!/usr/bin/env bash
# round 4: record every kernel description the GPU suite asks for (tests/spec_manifest.txt; __graft_entry__.build()
# pre-builds them so that the suite's 'specialised' pass finds every object).  Run on the GPU box:
#   gpurun -- tools/gpu/r04_record_manifest.sh        -> gpurun_out/spec_manifest.txt ; then
#   sort -u gpurun_out/spec_manifest.txt > tests/spec_manifest.txt
set -u
mkdir -p gpurun_out
rm -f gpurun_out/spec_manifest.raw
JAXSIM_AMD_TEST_RECORD=1 JAXSIM_AMD_SPEC_RECORD=$PWD/gpurun_out/spec_manifest.raw timeout ${T:-900} python -m pytest tests -m gpu -q --durations=8 > gpurun_out/record_pytest.log 2>&1
echo "pytest rc=$?"; tail -15 gpurun_out/record_pytest.log
sort -u gpurun_out/spec_manifest.raw > gpurun_out/spec_manifest.txt; wc -l gpurun_out/spec_manifest.txt; rm -f gpurun_out/spec_manifest.raw
