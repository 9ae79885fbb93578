#!/bin/bash
# wide groups (two molecules per workgroup) against one molecule per workgroup at batch sizes around the round boundaries of 256 CUs
B="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-parity-gate --diffusion-steps 250"
for b in 384 512 640 768 896 1024 1280; do
  for p in 0 2 1; do
    GAUDI_PAIRS=$p $B --batch $b 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('B=$b GAUDI_PAIRS=$p', round(d['value']/4,1), 'mol/s', d['config'].get('workgroups_per_call'), 'workgroups of', d['config'].get('node_slots_per_workgroup'))"
  done
done
