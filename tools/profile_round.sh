#!/bin/bash
# Collect the rocprofv3 evidence of one kernel state on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r01e
# Writes gpurun_out/prof_<tag>/... ; tools/summarize_profiles.py <tag> then condenses it into profiles/.
# Counter passes are separate from each other and use --kernel-trace only (no sys/hip/hsa tracing with --pmc).
tag=${1:-r01x}
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
R="rocprofv3 --kernel-trace --stats --output-format csv"
$R -d $out/c3 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/c3_bench.json 2> $out/c3.log
$R -d $out/c2 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --workload c2 > $out/c2_bench.json 2> $out/c2.log
$R -d $out/stab -- python3 bench.py --workload stability --steps 20 --warmup 1 --no-cpu-baseline > $out/stab_bench.json 2> $out/stab.log
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -- python3 bench.py --steps 1 --warmup 0 \
    --no-cpu-baseline --diffusion-steps 100 > $out/pmc_$c.json 2> $out/pmc_$c.log
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_c2_$c -- python3 bench.py --steps 1 --warmup 0 \
    --no-cpu-baseline --diffusion-steps 100 --workload c2 > $out/pmc_c2_$c.json 2> $out/pmc_c2_$c.log
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU \
  --kernel-trace --output-format csv -d $out/pmc_sq -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline \
  --diffusion-steps 100 > $out/pmc_sq.json 2> $out/pmc_sq.log
find $out -name "*.csv" | head -40
