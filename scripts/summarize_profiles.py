#!/usr/bin/env python3
"""Turns the rocprofv3 output of scripts/collect_profiles.sh (under gpurun_out/) into the committed summaries:

  profiles/<tag>_<cfg>_kernel_stats.csv           rocprofv3 --kernel-trace --stats, 4 streams (as in bench.py's timed region)
  profiles/<tag>_<cfg>_kernel_stats_streams1.csv  the same with --streams 1: exclusive kernel durations
  profiles/<tag>_<cfg>_pmc.csv                    per kernel: FETCH_SIZE / WRITE_SIZE (own passes) and the SQ counters
  profiles/<tag>_<cfg>_bench_streams1.json        the bench line printed under the profiler (--streams 1)
  profiles/pmc_traffic.json                       HBM bytes per launch, read by bench.py ("roofline.traffic")

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 64 B per 128-B request for coalesced
streaming reads, i.e. half the bytes; the factor was calibrated in round 1 on k_pack, whose read size is known
exactly (8 B per lane, coalesced -- the access shape of the solver kernels): 1.976.  WRITE_SIZE is exact.  Both in KiB.
FETCH_SIZE appears to include Infinity-Cache hits: for the fused kernel, whose per-instance state is re-read from the
L2 / Infinity Cache every pass, it is an upper bound of the HBM traffic.
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
cfg = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
FETCH_CORR = 1.976
go = os.path.join(ROOT, "gpurun_out")
pdir = os.path.join(ROOT, "profiles")
os.makedirs(pdir, exist_ok=True)


def short(name):
    name = name.replace("void ", "")
    base = name.split("<")[0].split("(")[0].replace("rmpc::", "")
    if base == "k_riccati":   # the two instantiations launched per pass
        base += "_grouped" if ", 4>" in name else "_1wave"
    return base


def newest(pattern):
    return max(glob.glob(pattern, recursive=True), key=os.path.getmtime)


for suffix, out in (("s4", "kernel_stats"), ("s1", "kernel_stats_streams1")):
    stats = newest(os.path.join(go, f"prof_{tag}_{cfg}_{suffix}", "**", "*kernel_stats.csv"))
    rows = [r for r in csv.DictReader(open(stats)) if "rmpc::" in r["Name"]]
    with open(os.path.join(pdir, f"{tag}_{cfg}_{out}.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows:
            w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])
    print(open(os.path.join(pdir, f"{tag}_{cfg}_{out}.csv")).read())

counters = collections.defaultdict(lambda: collections.defaultdict(list))
# (gpurun merges a call's files into gpurun_out/ without removing those of earlier calls: per counter pass only the
#  newest collection counts, or the means would mix two libraries)
for d in sorted(glob.glob(os.path.join(go, f"prof_{tag}_{cfg}_pmc*"))):
    if not os.path.isdir(d):
        continue
    f = newest(os.path.join(d, "**", "*counter_collection.csv"))
    for r in csv.DictReader(open(f)):
        if "rmpc::" in r["Kernel_Name"]:
            counters[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in counters.values() for c in k})
traffic = {}
with open(os.path.join(pdir, f"{tag}_{cfg}_pmc.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "launches"] + [n + "_mean" for n in names] + ["hbm_bytes_per_launch_mean", "hbm_bytes_per_launch_max"])
    for k in sorted(counters):
        c = counters[k]
        n = max(len(v) for v in c.values())
        fv, wv = c.get("FETCH_SIZE", [0.0]), c.get("WRITE_SIZE", [0.0])
        mean_b = (FETCH_CORR * sum(fv) / len(fv) + sum(wv) / len(wv)) * 1024.0
        max_b = (FETCH_CORR * max(fv) + max(wv)) * 1024.0
        w.writerow([k, n] + [f"{sum(c[m]) / len(c[m]):.6g}" if m in c else "" for m in names] + [f"{mean_b:.0f}", f"{max_b:.0f}"])
        traffic[k] = mean_b
        traffic[k + "_max"] = max_b
print(open(os.path.join(pdir, f"{tag}_{cfg}_pmc.csv")).read())
tfile = os.path.join(pdir, "pmc_traffic.json")
allt = json.load(open(tfile)) if os.path.exists(tfile) else {}
allt[cfg] = traffic
# (the library the passes ran: the bench line printed under the profiler carries its source hash -- bench.py reports
#  `traffic` only for that library)
lib_hash = None
try:
    lib_hash = json.loads([l for l in open(os.path.join(go, f"prof_{tag}_{cfg}_s1.json")) if l.startswith("{")][-1]).get("library_source_hash")
except Exception:   # noqa
    pass
allt.setdefault("_meta", {})[cfg] = {"tag": tag, "fetch_correction": FETCH_CORR, "library_source_hash": lib_hash,
                                     "source": f"profiles/{tag}_{cfg}_pmc.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, "
                                               "bench.py --streams 1; FETCH_SIZE x 1.976 + WRITE_SIZE, KiB -> bytes, mean over the launches)"}
json.dump(allt, open(tfile, "w"), indent=1, sort_keys=True)
try:
    line = [l for l in open(os.path.join(go, f"prof_{tag}_{cfg}_s1.json")) if l.startswith("{")][-1]
    json.dump(json.loads(line), open(os.path.join(pdir, f"{tag}_{cfg}_bench_streams1.json"), "w"), indent=1)
except Exception as e:   # noqa
    print("no bench line:", e)
