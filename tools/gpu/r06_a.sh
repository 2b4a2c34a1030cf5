#!/usr/bin/env bash
# round 6, call A: the new system_dynamics / link_contact_forces entries through the library's kernels, then the whole GPU
# suite under the library policy (regression check of the early-clobber asm operands and the new KArgs layout), then a bench line
set -u
OUT=gpurun_out/r06_a
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "(system_dynamics or link_contact or link_forces_from) and not specialised" > $OUT/pytest_dyn.log 2>&1; echo "pytest dyn rc=$?"; tail -5 $OUT/pytest_dyn.log
timeout 1500 python -m pytest tests/ -q -m gpu -k "not specialised" -x --deselect tests/test_specialize.py --deselect tests/test_bench_gpu.py > $OUT/pytest_lib.log 2>&1; echo "pytest lib rc=$?"; tail -5 $OUT/pytest_lib.log
JAXSIM_AMD_SPECIALIZE=0 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-other-contact-models --no-python-loop --saturated-envs 0 > $OUT/bench_lib.json 2> $OUT/bench_lib.err; echo "bench rc=$?"; tail -c 600 $OUT/bench_lib.json
