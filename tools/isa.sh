#!/bin/bash
# tools/isa.sh <kernel TU name without .hip> <tag> [extra hipcc flags] -> /tmp/isa/<name>_<tag>.s + instruction census
set -e
k=$1; tag=$2; shift 2
mkdir -p /tmp/isa
cd /root/repo/gaudi_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -S --cuda-device-only "$@" $k.hip -o /tmp/isa/${k}_${tag}.s 2>&1 | grep -v hip-link || true
f=/tmp/isa/${k}_${tag}.s
echo "$k [$tag] flat: $(grep -cE '^\s+flat_' $f) scratch: $(grep -cE '^\s+scratch_' $f) global: $(grep -cE '^\s+global_(load|store)_dw' $f) buffer_load: $(grep -cE '^\s+buffer_load' $f) mfma_f32x4: $(grep -c 'v_mfma_f32_16x16x4_f32' $f) mfma_bf16: $(grep -c 'v_mfma_f32_16x16x32_bf16' $f) mfma_f16: $(grep -c 'v_mfma_f32_16x16x32_f16' $f)"
grep -E '^\s+\.(vgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size):' $f | tr -s ' \t' ' ' | paste -sd' ' | fold -w 200
