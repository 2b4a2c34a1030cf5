#!/usr/bin/env bash
# round 3: per-phase cycle stamps of the specialised step kernel (profiling build of the spec object, built on the box)
set -u
export JAXSIM_AMD_SPEC_EXTRA_FLAGS="-DJXS_PHASE_TIMING ${EXTRA:-}"
JXS_DUO=0 JAXSIM_AMD_SPECIALIZE=1 timeout 300 python tools/phase_timing.py 2>&1 | head -24
