#!/usr/bin/env bash
# Build the oracle's C port (CPU baseline + cross-check of the NumPy oracle). gcc only.
set -euo pipefail
cd "$(dirname "$0")"
CC=${CC:-gcc}
FLAGS="-O3 -march=native -fopenmp -fPIC -std=gnu11 -Wall -Wno-unused-function"
$CC $FLAGS -DREAL=double -DSUFFIX=f64 -c step_ref.c -o step_ref_f64.o
$CC $FLAGS -DREAL=float -DSUFFIX=f32 -c step_ref.c -o step_ref_f32.o
$CC -shared -fopenmp step_ref_f64.o step_ref_f32.o -o liboracle_cport.so -lm
echo "built $(pwd)/liboracle_cport.so"
