#!/bin/bash
# Build an experimental variant of the two production kernels into diag/libgaudi_<name>.so
#   tools/build_variant.sh <name> [extra -D flags...]
set -e
name=$1; shift
cd "$(dirname "$0")/../gaudi_amd/csrc"
mkdir -p ../../diag
timeout 900 /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -shared -DGAUDI_STAMP_STUBS "$@" \
  -o ../../diag/libgaudi_$name.so gaudi_hip.hip kern_edm_192.hip kern_fused_192_208.hip kern8_edm_192.hip kern8_fused_192_208.hip
