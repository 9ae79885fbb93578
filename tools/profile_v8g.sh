#!/bin/bash
# rocprofv3 evidence for the V8G kernels (BASELINE config 4 read literally: bench.py --workload c4x) on the GPU box:
#   tools/profile_v8g.sh r04w   -> gpurun_out/prof_<tag>/c4x*, condensed into gpurun_out/prof_<tag>/summary/ (copy to profiles/)
# Counter passes are separate from each other and use --kernel-trace only (no sys/hip/hsa tracing with --pmc).
tag=${1:-r04w}
out=gpurun_out/prof_$tag
mkdir -p $out/summary
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
B="python3 bench.py --workload c4x --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --no-parity-gate"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/c4x -- $B > $out/c4x_bench.json 2> $out/c4x.log
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 \
  --kernel-trace --output-format csv -d $out/pmc_c4x_sq1 -- $B --diffusion-steps 100 > $out/pmc_c4x_sq1.json 2> $out/pmc_c4x_sq1.log
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_FLAT \
  --kernel-trace --output-format csv -d $out/pmc_c4x_sq2 -- $B --diffusion-steps 100 > $out/pmc_c4x_sq2.json 2> $out/pmc_c4x_sq2.log
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_FLAT SQ_INSTS_SALU SQ_INSTS_SMEM \
  --kernel-trace --output-format csv -d $out/pmc_c4x_sq3 -- $B --diffusion-steps 100 > $out/pmc_c4x_sq3.json 2> $out/pmc_c4x_sq3.log
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_c4x_$c -- $B --diffusion-steps 100 > $out/pmc_c4x_$c.json 2> $out/pmc_c4x_$c.log
done
python3 - "$out" "$tag" <<'PY'
import csv, glob, os, shutil, sys
out, tag = sys.argv[1], sys.argv[2]
hits = glob.glob(os.path.join(out, "c4x/**/*kernel_stats.csv"), recursive=True)
if hits: shutil.copy(hits[0], os.path.join(out, "summary", f"{tag}_c4x_kernel_stats.csv"))
if os.path.exists(os.path.join(out, "c4x_bench.json")): shutil.copy(os.path.join(out, "c4x_bench.json"), os.path.join(out, "summary", f"{tag}_c4x_bench_under_rocprof.json"))
rows = []
for d in sorted(glob.glob(os.path.join(out, "pmc_c4x_*"))):
    if not os.path.isdir(d): continue
    f = glob.glob(os.path.join(d, "**/*counter_collection.csv"), recursive=True)
    if not f: continue
    per = {}
    for r in csv.DictReader(open(f[0])):
        if "sampler_kernel" not in r["Kernel_Name"]: continue
        v = per.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], [0.0, 0.0])
        v[0] += float(r["Counter_Value"])
        if r.get("End_Timestamp"): v[1] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    for key, disp in per.items():
        vals = [v[0] for v in disp.values()]; durs = [v[1] for v in disp.values()]
        rows.append(dict(workload="c4x", rocprofv3_pass=os.path.basename(d), counter=key, launches=len(vals),
                         mean_value_per_launch=sum(vals) / len(vals), mean_launch_ns=sum(durs) / max(len(durs), 1)))
if rows:
    with open(os.path.join(out, "summary", f"{tag}_c4x_pmc_summary.csv"), "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
print("summary:", os.listdir(os.path.join(out, "summary")))
PY
