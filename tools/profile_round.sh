#!/usr/bin/env bash
# Collect the rocprofv3 evidence of one round on the GPU box (run from the repo root through gpurun):
#   tools/profile_round.sh <run-name>      -> gpurun_out/<run-name>/{stats,pmc_*,c5,...}
# then, back in the container:  python tools/summarize_profiles.py gpurun_out/<run-name> r01
# Counter passes are separate runs (`--pmc` is never combined with sys/hip/hsa tracing).
set -u
R=$PWD
OUT=$R/gpurun_out/$1
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 500 --warmup 20 --no-cpu-baseline --saturated-envs 0 --no-other-contact-models --no-python-loop"
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/stats" -- $B > "$OUT/bench.log" 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -- $B > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmc_write" -- $B > "$OUT/pmc_write.log" 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY \
  -d "$OUT/pmc_sq" -- $B > "$OUT/pmc_sq.log" 2>&1
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/c5" -- python "$R/tools/bench_c5.py" > "$OUT/c5.log" 2>&1
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/c5_relaxed" -- python "$R/tools/bench_c5.py" --contact relaxed --points 16 > "$OUT/c5_relaxed.log" 2>&1
# the reference's own step-benchmark idiom: humanoid, all 32 points, RelaxedRigidContacts with estimated parameters (link space, HISTORY.md 4m)
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/relaxed_humanoid" -- python "$R/tools/bench_c5.py" --contact relaxed --points 32 --envs 1024 > "$OUT/relaxed_humanoid.log" 2>&1
cd "$R"
python bench.py > "$OUT/bench_N1.json" 2> "$OUT/bench_N1.err"
python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_steps20.json" 2> "$OUT/bench_steps20.err"
# cycle stamps: the model-specialised kernels with -DJXS_PHASE_TIMING (built on first use, seconds), then the generic
# kernels of a timing build of the library (cd jaxsim_amd/csrc && JXS_EXTRA_FLAGS=-DJXS_PHASE_TIMING JXS_OUT=libjaxsim_amd_timing.so bash build.sh)
export JAXSIM_AMD_SPEC_EXTRA_FLAGS=-DJXS_PHASE_TIMING
JAXSIM_AMD_SPECIALIZE=1 python tools/phase_timing.py > "$OUT/phases.log" 2>&1
# (the two-wave experiment of round 3 is closed: profiles/r03_two_wave_experiment.md; the variant lives in the library only)
for a in "4 4096" "4 4096 rigid standing" "16 4096" "16 4096 relaxed standing" "32 1024 relaxed standing"; do
  JAXSIM_AMD_SPECIALIZE=1 python tools/phase_timing_rigid.py $a >> "$OUT/phases_contact_models.log" 2>&1
done
unset JAXSIM_AMD_SPEC_EXTRA_FLAGS
if [ -f jaxsim_amd/csrc/libjaxsim_amd_timing.so ]; then
  JAXSIM_AMD_SPECIALIZE=0 JAXSIM_AMD_LIB=$R/jaxsim_amd/csrc/libjaxsim_amd_timing.so python tools/phase_timing.py > "$OUT/phases_generic.log" 2>&1
  JAXSIM_AMD_SPECIALIZE=0 JAXSIM_AMD_LIB=$R/jaxsim_amd/csrc/libjaxsim_amd_timing.so python tools/phase_timing_rigid.py 4 4096 > "$OUT/phases_contact_models_generic.log" 2>&1
fi
python tools/fp32_error_gpu.py 512 > "$OUT/fp32_error_gpu.log" 2>&1
timeout 60 tools/ubench/issue_rate > "$OUT/issue_rate.log" 2>&1
timeout 120 tools/ubench/cu_share > "$OUT/cu_share.log" 2>&1
timeout 300 python tools/sweep.py --sizes 1024,2048,4096,65536 --steps 1000 >> "$OUT/sweep.log" 2>&1
# PMC pass for the config-5 kernel (HBM-side traffic of the rigid-contact step)
cd /tmp
BC="python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --saturated-envs 0"   # (with the secondary contact-model figures: config 5 runs inside)
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d "$OUT/c5_pmc_fetch" -- $BC > "$OUT/c5_pmc_fetch.log" 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d "$OUT/c5_pmc_write" -- $BC > "$OUT/c5_pmc_write.log" 2>&1
cd "$R"
tail -c 600 "$OUT/bench_N1.json"
# summarise on the box (the raw per-dispatch CSVs are tens of MB; only gpurun_out/ <= 64 MiB travels back), keep the
# kernel statistics, drop the raw traces
python tools/summarize_profiles.py "gpurun_out/$1" "${2:-r03}" "$OUT/profiles" > "$OUT/summarize.log" 2>&1
# the bench line ties `roofline.traffic` to the PMC record of the SAME kernel sources: put the fresh record where bench.py
# looks for it, run the two bench commands again and summarise again (so that the committed bench lines carry the traffic)
cp "$OUT/profiles/${2:-r03}_pmc.json" "$R/profiles/" 2>/dev/null
python bench.py > "$OUT/bench_N1.json" 2> "$OUT/bench_N1.err"
python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_steps20.json" 2> "$OUT/bench_steps20.err"
python tools/summarize_profiles.py "gpurun_out/$1" "${2:-r03}" "$OUT/profiles" > "$OUT/summarize.log" 2>&1
find "$OUT" -name "*counter_collection.csv" -delete
find "$OUT" -name "*kernel_trace.csv" -delete
find "$OUT" -name "*.db" -delete
du -sh "$OUT"
