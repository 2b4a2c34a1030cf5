#!/usr/bin/env bash
# Build libjaxsim_amd.so for gfx950 (cross-compiles without a GPU).
#   build.sh                 -> libjaxsim_amd.so
#   JXS_EXTRA_FLAGS=-DJXS_PHASE_TIMING JXS_OUT=libjaxsim_amd_timing.so build.sh   (developer profiling build)
#   JXS_ONLY="float:0 float:6" build.sh   (developer: recompile only these dtype:mode units; the other
#                                          objects of the build directory are reused)
# The kernels of one (dtype, mode) pair are one translation unit (jxs_inst.hip); the 28 units and the C-ABI
# file compile in parallel.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=${JXS_OUT:-libjaxsim_amd.so}
OBJ=build/${OUT%.so}
mkdir -p "$OBJ"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -mllvm -amdgpu-kernarg-preload-count=16 -Wall -Wno-unused-function -Wno-cuda-compat -DJXS_WITH_DUO ${JXS_EXTRA_FLAGS:-}"
JOBS=${JXS_JOBS:-$(nproc)}
UNITS=${JXS_ONLY:-}
if [ -z "$UNITS" ]; then
  for t in float double; do for m in 0 1 2 3 4 5 6 7 8 9 10 11 12 13; do UNITS="$UNITS $t:$m"; done; done
  UNITS="$UNITS api"
fi
compile() {
  local u=$1 o
  if [ "$u" = api ]; then
    o="$OBJ/api.o"; rm -f "$o"
    "$HIPCC" $FLAGS -c jxs_api.hip -o "$o" || { echo "$u" >> "$OBJ/failed"; return 1; }
  else
    local t=${u%%:*} m=${u##*:}
    o="$OBJ/inst_${t}_${m}.o"; rm -f "$o"     # a stale object must never be linked after a failed compile
    "$HIPCC" $FLAGS -DJXS_INST_T=$t -DJXS_INST_MODE=$m -c jxs_inst.hip -o "$o" || { echo "$u" >> "$OBJ/failed"; return 1; }
  fi
}
export -f compile
export HIPCC FLAGS OBJ
# the slowest units first (rigid contact modes 6, 7; Runge-Kutta 5)
ORDERED=$(for u in $UNITS; do case $u in *:7) echo "0 $u";; *:6|*:13) echo "1 $u";; *:5) echo "2 $u";; *) echo "3 $u";; esac; done | sort -s -k1,1 | cut -d' ' -f2)
rm -f "$OBJ/failed" "$OBJ/compile.log"
set +e
printf '%s\n' $ORDERED | xargs -P "$JOBS" -I{} bash -c 'compile {}' > "$OBJ/compile.log" 2>&1
RC=$?
set -e
grep -v "warning: loop not unrolled\|^ *[0-9]* *|\|\^\|warning generated\|warnings generated" "$OBJ/compile.log" || true
if [ $RC -ne 0 ] || [ -s "$OBJ/failed" ]; then
  echo "build.sh: compilation failed (xargs rc $RC) for units: $(tr '\n' ' ' < "$OBJ/failed" 2>/dev/null)" >&2
  exit 1
fi
for t in float double; do for m in 0 1 2 3 4 5 6 7 8 9 10 11 12 13; do [ -f "$OBJ/inst_${t}_${m}.o" ] || { echo "missing object inst_${t}_${m}.o" >&2; exit 1; }; done; done
[ -f "$OBJ/api.o" ] || { echo "missing object api.o" >&2; exit 1; }
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "$OBJ"/api.o "$OBJ"/inst_*.o -o "$OUT" -ldl
# [round 4] wait-state lint of the device code (jaxsim_amd/isa_lint.py): the hazards hipcc cannot pad inside asm blocks
# (VALU write -> DPP read, VALU EXEC write -> DPP).  A hit FAILS the build.  JXS_SKIP_LINT=1: developer builds only.
if [ "${JXS_SKIP_LINT:-0}" != 1 ]; then
  (cd ../.. && "${PYTHON:-python3}" -m jaxsim_amd.isa_lint "jaxsim_amd/csrc/$OUT") || { echo "build.sh: ISA lint failed for $OUT" >&2; rm -f "$OUT"; exit 1; }
fi
echo "built $(pwd)/$OUT"
