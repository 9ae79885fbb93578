B="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-parity-gate --batch 1024 --nodes 9 --diffusion-steps 250"
for rep in 1 2; do
  for fh in 0 1; do
    if [ $fh = 1 ]; then export GAUDI_FORCE_HALF=1; else unset GAUDI_FORCE_HALF; fi
    GAUDI_DEBUG_PLAN=1 $B 2>gpurun_out/ringab_$fh.err | python3 -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('force_half=$fh', d['value'], d['config'].get('workgroups_per_call'), d['config'].get('node_slots_per_workgroup'))"
    grep plan gpurun_out/ringab_$fh.err | tail -1
  done
done
