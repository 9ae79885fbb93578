#!/bin/bash
# Register / scratch usage of the production kernels and their out-of-line phases, from hipcc's own accounting:
#   tools/resource_usage.sh > profiles/<tag>_resource_usage.txt
cd "$(dirname "$0")/../gaudi_amd/csrc"
for tu in kern8s_fused_192_208 kern8s2_fused_192_208 kern8s_edm_192 kern8m_fused_192_208_h kern8g_fused_192_208 kern8h_fused_192_208 kern8_fused_192_208 kern8_edm_192 kern_fused_192_208 kern_edm_192; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -S --cuda-device-only $tu.hip -o /tmp/ru_$tu.s 2>/dev/null
  python3 - /tmp/ru_$tu.s $tu <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
print(f"== {sys.argv[2]}.hip")
for f in re.split(r'\n(?=_ZN[^\n]*:\s*;? ?@?)', txt):
    name = f.split('\n', 1)[0]
    if not name.startswith('_ZN'):
        continue
    short = re.sub(r'^_ZN5gaudi\d+', '', name.split(':')[0])[:48]
    lines = f.split('\n')
    get = lambda key: next((l.split(':')[1].strip() for l in lines if l.startswith('; ' + key + ':')), '?')
    sc = [k for k, l in enumerate(lines) if 'scratch_' in l]
    mid = sum(1 for k in sc if 300 < k < len(lines) - 300)
    print(f"  {short:50s} VGPRs {get('NumVgprs'):>4s}  AGPRs {get('NumAgprs'):>4s}  SGPRs {get('NumSgprs'):>4s}  scratch {get('ScratchSize'):>5s} B/lane  "
          f"scratch instructions {len(sc):4d} ({mid} outside prologue/epilogue)  MFMA {sum('v_mfma' in l for l in lines):5d}  "
          f"sgpr-spill lanes (v_writelane) {sum('v_writelane' in l for l in lines):4d}")
PY
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -c $tu.hip -o /tmp/ru.o -Rpass-analysis=kernel-resource-usage 2>&1 | \
    grep -E "SGPRs Spill|VGPRs Spill|Occupancy" | sed 's/.*remark: *//; s/ \[-R.*//' | tr '\n' ';'; echo
done
