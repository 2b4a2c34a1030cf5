#!/usr/bin/env bash
# round 5, call H: record the kernel descriptions of the GPU suite again (new tests: the 200-point quadruped), and the
# figure of that model
set -u
OUT=gpurun_out/r05_h
mkdir -p $OUT
T=1200 bash tools/gpu/r04_record_manifest.sh
for st in "" "--standing"; do
  JAXSIM_AMD_SPECIALIZE=1 timeout 600 python tools/bench_c5.py --contact relaxed --points 200 $st >> $OUT/c5_200.txt 2>> $OUT/err.log
done
cat $OUT/c5_200.txt
