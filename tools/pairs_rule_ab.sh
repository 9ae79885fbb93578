#!/bin/bash
# wide groups (two molecules per workgroup) against one molecule per workgroup at batch sizes around the round boundaries of 256 CUs:
# GAUDI_PAIRS=0 one per workgroup, 2 two per workgroup always, 1 the default (a mixed launch where that is the shortest schedule)
B="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-parity-gate --diffusion-steps 250"
run() { env "$@" $B --batch $b 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('B=$b $*', round(d['value']/4,1), 'mol/s', d['config'].get('workgroups_per_call'), 'workgroups of', d['config'].get('node_slots_per_workgroup'))"; }
b=512; run GAUDI_PAIRS=1 GAUDI_PAIRS_CAP=1
for b in 384 512 640 768 896 1024 1280; do
  run GAUDI_PAIRS=0; run GAUDI_PAIRS=2; run GAUDI_PAIRS=1
done
