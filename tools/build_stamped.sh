#!/bin/bash
# Diagnostic build with in-kernel phase stamps (shares only; never used for reported timings).
#   tools/build_stamped.sh [g] && GAUDI_LIB=$PWD/gaudi_amd/libgaudi_hip_stamps.so GAUDI_PRINT_STAMPS=1 python bench.py ...
#   g: also the fused V8G kernel (node buffers in global memory) -> c4x / GAUDI_FORCE_GN8=1 runs
#   m: also the fused MR half-ring kernel -> wide groups (--batch 1024, GAUDI_PAIRS=1)
set -e
EXTRA=""
if [ "$1" = "g" ]; then EXTRA="kern8g_fused_192_208.hip -DGAUDI_STAMP_G"; fi
if [ "$1" = "m" ]; then EXTRA="kern8m_fused_192_208_h.hip -DGAUDI_STAMP_M"; fi  # the MR half-ring kernel: wide groups (GAUDI_PAIRS)
cd "$(dirname "$0")/../gaudi_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -shared -DGAUDI_STAMPS \
  -o ../libgaudi_hip_stamps.so gaudi_hip.hip kern_edm_192.hip kern_fused_192_208.hip kern8_edm_192.hip kern8_fused_192_208.hip kern8s_edm_192.hip kern8s_fused_192_208.hip kern8h_fused_192_208.hip \
  -DGAUDI_STAMP_STUBS $EXTRA $GAUDI_EXTRA_FLAGS
