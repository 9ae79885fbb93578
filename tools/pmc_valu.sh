#!/bin/bash
# VALU instruction count of one C3 / C2 launch set (separate --pmc pass, kernel trace only): tools/pmc_valu.sh <tag>
tag=${1:-x}
out=gpurun_out/pmc_valu_$tag
mkdir -p $out
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
B="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --no-parity-gate --diffusion-steps 100"
for wl in c3 c2; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $out/$wl -- $B --workload $wl > $out/$wl.json 2> $out/$wl.log
done
python3 - $out <<'PY'
import csv, glob, sys, collections
for wl in ("c3", "c2"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{sys.argv[1]}/{wl}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "sampler_kernel" in r["Kernel_Name"]:
                acc[(r["Dispatch_Id"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    tot = collections.defaultdict(list)
    for (d, c), v in acc.items():
        tot[c].append(sum(v))
    print(wl, {c: f"{sum(v)/len(v):.4g}" for c, v in sorted(tot.items())})
PY
