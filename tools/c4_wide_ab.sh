#!/bin/bash
# C4 (hetero 3-10 rings, B = 1024) with the default packing against forced wide groups (GAUDI_PAIRS=2), half ring and full ring
B="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --workload c4 --diffusion-steps 250"
for rep in 1 2; do
  for cfg in "default:" "wide_full:GAUDI_PAIRS=2" "wide_half:GAUDI_PAIRS=2 GAUDI_WIDE_FULL=0"; do
    name=${cfg%%:*}; envs=${cfg#*:}
    env $envs GAUDI_DEBUG_PLAN=1 $B 2>gpurun_out/c4w_$name.err | python3 -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$name', round(d['value']/4,1), d['config'].get('workgroups_per_call'), d['config'].get('node_slots_per_workgroup'), d['parity_gate']['rel_err'])"
    grep plan gpurun_out/c4w_$name.err | tail -1
  done
done
