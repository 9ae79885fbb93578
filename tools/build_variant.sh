#!/bin/bash
# Build an experimental variant of the production kernels into gaudi_amd/libgaudi_var_<name>.so (git-ignored, travels to the GPU box)
#   tools/build_variant.sh <name> [extra -D flags...]
set -e
name=$1; shift
cd "$(dirname "$0")/../gaudi_amd/csrc"
timeout 900 /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -shared -DGAUDI_STAMP_STUBS "$@" \
  -o ../libgaudi_var_$name.so gaudi_hip.hip kern_edm_192.hip kern_fused_192_208.hip kern8_edm_192.hip kern8_fused_192_208.hip \
  kern8s_edm_192.hip kern8s_fused_192_208.hip kern8h_fused_192_208.hip
