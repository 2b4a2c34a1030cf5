#!/usr/bin/env bash
# Build libjaxsim_amd.so for gfx950 (cross-compiles without a GPU).
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-slp-vectorize -mllvm -amdgpu-kernarg-preload-count=16 -Wall -Wno-unused-function \
  ${JXS_EXTRA_FLAGS:-} jxs_api.hip -o libjaxsim_amd.so -ldl
echo "built $(pwd)/libjaxsim_amd.so"
