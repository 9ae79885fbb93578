#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ (written by tools/profile_round.sh on the GPU box) into profiles/:
<tag>_{c3,c2,stability}_kernel_stats.csv (rocprofv3 --stats tables), <tag>_pmc_summary.csv (mean counter value and
launch duration per sampler/stability kernel launch) and pmc_traffic.json (HBM-side bytes per launch for bench.py)."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")


def one(pattern):
    hits = glob.glob(os.path.join(src, pattern), recursive=True)
    return hits[0] if hits else None


for wl in ("c3", "c2", "c4", "c3fp32", "stab"):
    f = one(f"{wl}/**/*kernel_stats.csv")
    if f:
        shutil.copy(f, os.path.join(dst, f"{tag}_{'stability' if wl == 'stab' else wl}_kernel_stats.csv"))
        print("copied", f)
    b = os.path.join(src, f"{wl}_bench.json")
    if os.path.exists(b) and os.path.getsize(b):
        shutil.copy(b, os.path.join(dst, f"{tag}_{'stability' if wl == 'stab' else wl}_bench_under_rocprof.json"))

rows = []
traffic = {}
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    name = os.path.basename(d)
    wl = "c2" if "_c2_" in name else "c4" if "_c4_" in name else "c4x" if "_c4x_" in name else "c3"
    f = one(f"{name}/**/*counter_collection.csv")
    if not f:
        continue
    per = {}
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if "sampler_kernel" not in r["Kernel_Name"]:
                continue
            key = r["Counter_Name"]
            per.setdefault(key, {}).setdefault(r["Dispatch_Id"], [0.0, 0.0])
            per[key][r["Dispatch_Id"]][0] += float(r["Counter_Value"])
            if "End_Timestamp" in r and r["End_Timestamp"]:
                per[key][r["Dispatch_Id"]][1] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    for key, disp in per.items():
        vals = [v[0] for v in disp.values()]
        durs = [v[1] for v in disp.values()]
        # launches of 25 steps only (the last launch of a chain also decodes; same order of magnitude)
        rows.append(dict(workload=wl, rocprofv3_pass=name, counter=key, launches=len(vals),
                         mean_value_per_launch=sum(vals) / len(vals), mean_launch_ns=sum(durs) / max(len(durs), 1)))
        if key in ("FETCH_SIZE", "WRITE_SIZE"):
            traffic.setdefault(wl, {})[key] = sum(vals) / len(vals)
with open(os.path.join(dst, f"{tag}_pmc_summary.csv"), "w", newline="") as fh:
    w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
    w.writeheader()
    w.writerows(rows)
# round 4: the wide-group passes (C3 at 1024 molecules, GAUDI_PAIRS = 0 / 1), summarised on their own
wrows = []
for d in sorted(glob.glob(os.path.join(src, "wide_pairs*_sq*"))):
    if not os.path.isdir(d):
        continue
    name = os.path.basename(d)
    f = one(f"{name}/**/*counter_collection.csv")
    if not f:
        continue
    per = {}
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if "sampler_kernel" not in r["Kernel_Name"]:
                continue
            per.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], [0.0, 0.0])
            per[r["Counter_Name"]][r["Dispatch_Id"]][0] += float(r["Counter_Value"])
            if "End_Timestamp" in r and r["End_Timestamp"]:
                per[r["Counter_Name"]][r["Dispatch_Id"]][1] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    for key, disp in per.items():
        vals, durs = [v[0] for v in disp.values()], [v[1] for v in disp.values()]
        wrows.append(dict(launch="GAUDI_PAIRS=" + name.split("pairs")[1][0], rocprofv3_pass=name, counter=key, launches=len(vals),
                          mean_value_per_launch=sum(vals) / len(vals), mean_launch_ns=sum(durs) / max(len(durs), 1)))
if wrows:
    with open(os.path.join(dst, f"{tag}_wide_groups_pmc.csv"), "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(wrows[0].keys()))
        w.writeheader()
        w.writerows(wrows)
    for pr in "01":
        f = one(f"wide_pairs{pr}_stats/**/*kernel_stats.csv")
        if f:
            shutil.copy(f, os.path.join(dst, f"{tag}_wide_pairs{pr}_kernel_stats.csv"))
        b = os.path.join(src, f"wide_pairs{pr}_bench.json")
        if os.path.exists(b) and os.path.getsize(b):
            shutil.copy(b, os.path.join(dst, f"{tag}_wide_pairs{pr}_bench_under_rocprof.json"))
out = {}
for wl, t in traffic.items():
    if "FETCH_SIZE" in t and "WRITE_SIZE" in t:
        out[f"{wl}_bytes_per_launch_raw"] = (t["FETCH_SIZE"] + t["WRITE_SIZE"]) * 1024.0
        out[f"{wl}_bytes_per_launch"] = (2.0 * t["FETCH_SIZE"] + t["WRITE_SIZE"]) * 1024.0
out["note"] = (f"per 25-step launch, B=256 (c4: B=1024); FETCH_SIZE x2 (MI355X_MICROARCH.md: gfx950 reports 1/2 of 16 B/lane coalesced "
               f"reads) + WRITE_SIZE, KiB->bytes; kernel state {tag}")
sys.path.insert(0, ROOT)
from gaudi_amd import build as _build  # noqa: E402
out["csrc_sha256"] = _build.csrc_digest()  # bench.py reports these figures only for the kernels they were measured on
if len(out) > 2:
    with open(os.path.join(dst, "pmc_traffic.json"), "w") as fh:
        json.dump(out, fh, indent=1)
print(json.dumps(out, indent=1))
for r in rows:
    print(r)
