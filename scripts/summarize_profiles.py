#!/usr/bin/env python3
"""Turns the rocprofv3 output of scripts/collect_profiles.sh (under gpurun_out/)
into the committed summaries under profiles/:

  profiles/<tag>_<cfg>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (rmpc kernels)
  profiles/<tag>_<cfg>_pmc.csv            FETCH_SIZE / WRITE_SIZE per kernel (separate passes)
  profiles/pmc_traffic.json               HBM bytes per launch, read by bench.py ("roofline.traffic")

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 64 B per
128-B request for coalesced streaming reads, i.e. half the bytes; calibrated here
on k_pack, whose read size is known exactly (8 B per lane, coalesced -- the same
access shape as the solver kernels).  WRITE_SIZE is exact.  Both are in KiB.
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
cfg = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
go = os.path.join(ROOT, "gpurun_out")
pdir = os.path.join(ROOT, "profiles")
os.makedirs(pdir, exist_ok=True)


def short(name):
    name = name.replace("void ", "")
    return name.split("<")[0].split("(")[0].replace("rmpc::", "")


def newest(pattern):
    # gpurun merges every call's files into gpurun_out/: take the latest collection
    return max(glob.glob(pattern), key=os.path.getmtime)


stats = newest(os.path.join(go, f"prof_{tag}", "*", "*kernel_stats.csv"))
rows = [r for r in csv.DictReader(open(stats)) if "rmpc::" in r["Name"]]
with open(os.path.join(pdir, f"{tag}_{cfg}_kernel_stats.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for r in rows:
        w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])

pmc = {}
for cname, d in (("FETCH_SIZE", f"pmc_fetch_{tag}"), ("WRITE_SIZE", f"pmc_write_{tag}")):
    f = newest(os.path.join(go, d, "*", "*counter_collection.csv"))
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == cname and "rmpc::" in r["Kernel_Name"]:
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    pmc[cname] = agg

# calibration of the read counter on k_pack (first launch = params array)
prof_json = json.load(open(os.path.join(go, f"prof_bench_{tag}.json")))
c = prof_json["config"]
known_pack_read = 8.0 * c["batch_per_gpu"] * c["horizon"] * c["npar"]
fetch_scale = known_pack_read / (max(pmc["FETCH_SIZE"]["k_pack"]) * 1024.0)

with open(os.path.join(pdir, f"{tag}_{cfg}_pmc.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "launches", "FETCH_SIZE_KiB_mean", "FETCH_SIZE_KiB_max", "WRITE_SIZE_KiB_mean", "WRITE_SIZE_KiB_max",
                "fetch_correction", "hbm_bytes_per_launch_mean", "hbm_bytes_per_launch_max"])
    traffic = {}
    for k in sorted(pmc["FETCH_SIZE"]):
        fv, wv = pmc["FETCH_SIZE"][k], pmc["WRITE_SIZE"].get(k, [0.0])
        mean_b = (fetch_scale * sum(fv) / len(fv) + sum(wv) / len(wv)) * 1024.0
        max_b = (fetch_scale * max(fv) + max(wv)) * 1024.0
        w.writerow([k, len(fv), f"{sum(fv)/len(fv):.1f}", f"{max(fv):.1f}", f"{sum(wv)/len(wv):.1f}", f"{max(wv):.1f}",
                    f"{fetch_scale:.3f}", f"{mean_b:.0f}", f"{max_b:.0f}"])
        traffic[k] = mean_b
        traffic[k + "_full_launch"] = max_b

tfile = os.path.join(pdir, "pmc_traffic.json")
allt = json.load(open(tfile)) if os.path.exists(tfile) else {}
allt[cfg] = traffic
allt.setdefault("_meta", {})[cfg] = {"tag": tag, "fetch_correction": fetch_scale,
                                     "source": f"profiles/{tag}_{cfg}_pmc.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"}
json.dump(allt, open(tfile, "w"), indent=1, sort_keys=True)
json.dump(prof_json, open(os.path.join(pdir, f"{tag}_{cfg}_bench_under_rocprof.json"), "w"), indent=1)
print(open(os.path.join(pdir, f"{tag}_{cfg}_kernel_stats.csv")).read())
print(open(os.path.join(pdir, f"{tag}_{cfg}_pmc.csv")).read())
