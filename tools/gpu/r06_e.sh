#!/usr/bin/env bash
# round 6, call E: TIMING experiment -- would a fourth resident wave per SIMD pay?  The specialised step kernel compiled for 128
# registers (JXS_MIN_WAVES=4, 88 bytes of scratch) with an LDS allocation of 10 KB per wave (the kernel uses 13.2: its accesses
# beyond the allocation are dropped, the results are garbage, the instruction stream is the same) against today's kernel.
set -u
OUT=gpurun_out/r06_e
mkdir -p $OUT
for fl in "" "-DJXS_MIN_WAVES=4" "-DJXS_MIN_WAVES=4 -DJXS_EXP_LDS_BYTES=10240" "-DJXS_EXP_LDS_BYTES=10240"; do
  echo "== flags: $fl" | tee -a $OUT/sweep.log
  JAXSIM_AMD_SPEC_EXTRA_FLAGS="$fl" JAXSIM_AMD_SPECIALIZE=1 python tools/sweep.py --sizes 1024,4096,6144,8192,12288,16384,65536 --steps 300 2>&1 | tee -a $OUT/sweep.log | cut -c40-120
done
