#!/bin/bash
# Collect the rocprofv3 evidence of one kernel state on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r02a
# Writes gpurun_out/prof_<tag>/... ; tools/summarize_profiles.py <tag> then condenses it into profiles/.
# Counter passes are separate from each other and use --kernel-trace only (no sys/hip/hsa tracing with --pmc).
tag=${1:-r02x}
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
R="rocprofv3 --kernel-trace --stats --output-format csv"
B="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --no-parity-gate"  # (the gate is one more, 1-step launch: kept out of the per-launch means)
$R -d $out/c3 -- $B > $out/c3_bench.json 2> $out/c3.log
$R -d $out/c2 -- $B --workload c2 > $out/c2_bench.json 2> $out/c2.log
$R -d $out/c4 -- $B --workload c4 > $out/c4_bench.json 2> $out/c4.log
GAUDI_EDGE_MATH=fp32 $R -d $out/c3fp32 -- $B > $out/c3fp32_bench.json 2> $out/c3fp32.log  # the fp32-instruction kernel beside it
$R -d $out/stab -- python3 bench.py --workload stability --steps 20 --warmup 1 --no-cpu-baseline > $out/stab_bench.json 2> $out/stab.log
for wl in c3 c2 c4; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_${wl}_$c -- $B --diffusion-steps 100 --workload $wl \
      > $out/pmc_${wl}_$c.json 2> $out/pmc_${wl}_$c.log
  done
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 \
    --kernel-trace --output-format csv -d $out/pmc_${wl}_sq1 -- $B --diffusion-steps 100 --workload $wl > $out/pmc_${wl}_sq1.json 2> $out/pmc_${wl}_sq1.log
  rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES \
    --kernel-trace --output-format csv -d $out/pmc_${wl}_sq2 -- $B --diffusion-steps 100 --workload $wl > $out/pmc_${wl}_sq2.json 2> $out/pmc_${wl}_sq2.log
  # round 5: the 16-bit matrix instructions are fp16 now (node and edge GEMMs); FLAT must stay at zero
  rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_FLAT SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS \
    --kernel-trace --output-format csv -d $out/pmc_${wl}_sq3 -- $B --diffusion-steps 100 --workload $wl > $out/pmc_${wl}_sq3.json 2> $out/pmc_${wl}_sq3.log
done
# round 4: wide groups (GAUDI_PAIRS=1: two cata molecules per workgroup at 1024 molecules) beside the default launch of the same batch
for pr in 0 1; do
  GAUDI_PAIRS=$pr $R -d $out/wide_pairs${pr}_stats -- $B --batch 1024 --diffusion-steps 100 > $out/wide_pairs${pr}_bench.json 2> $out/wide_pairs${pr}.log
  GAUDI_PAIRS=$pr rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 \
    --kernel-trace --output-format csv -d $out/wide_pairs${pr}_sq1 -- $B --batch 1024 --diffusion-steps 100 > $out/wide_pairs${pr}_sq1.json 2> $out/wide_pairs${pr}_sq1.log
  GAUDI_PAIRS=$pr rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES \
    --kernel-trace --output-format csv -d $out/wide_pairs${pr}_sq2 -- $B --batch 1024 --diffusion-steps 100 > $out/wide_pairs${pr}_sq2.json 2> $out/wide_pairs${pr}_sq2.log
done
find $out -name "*.csv" | wc -l
