#!/bin/bash
# PMC pass over a short C2 run for one kernel variant: tools/pmc_c2.sh <waves> <outdir> [workload]
w=$1; out=$2; wl=${3:-c2}
mkdir -p $out
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
export GAUDI_WAVES=$w
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_ACTIVE_INST_ANY \
  --kernel-trace --output-format csv -d $out/sq1 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --diffusion-steps 100 --workload $wl > $out/sq1.json 2> $out/sq1.log
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES \
  --kernel-trace --output-format csv -d $out/sq2 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --diffusion-steps 100 --workload $wl > $out/sq2.json 2> $out/sq2.log
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for d in ("sq1", "sq2"):
    for f in glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            if "sampler_kernel" in r["Kernel_Name"]:
                per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
        for k, v in per.items():
            vals = list(v.values())
            print(f"{d} {k:36s} launches={len(vals)} mean={sum(vals)/len(vals):.4g}")
PY
