#!/bin/bash
# how much of a workgroup's time depends on how many other workgroups run (chip-level contention): one molecule / two molecules per
# workgroup with half and all of the 256 CUs busy
B="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-parity-gate --diffusion-steps 250"
run() { bb=$1; shift; env "$@" GAUDI_DEBUG_PLAN=1 $B --batch $bb 2>/tmp/err.txt | python3 -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('B=$bb $*', round(d['ms_per_step']*4,1), 'ms per 1000 steps', d['roofline'].get('clock_mhz'))"; grep plan /tmp/err.txt | tail -1; }
run 64 GAUDI_PAIRS=0
run 128 GAUDI_PAIRS=0
run 256 GAUDI_PAIRS=0
run 128 GAUDI_PAIRS=2
run 256 GAUDI_PAIRS=2
run 512 GAUDI_PAIRS=2
run 256 GAUDI_PAIRS=2 GAUDI_WIDE_FULL=0
run 512 GAUDI_PAIRS=2 GAUDI_WIDE_FULL=0
