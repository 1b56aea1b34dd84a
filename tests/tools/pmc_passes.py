"""Per-dispatch PMC counters of the pass kernels from rocprofv3 counter_collection CSVs (development aid).
usage: pmc_passes.py <counter_collection.csv> [...]; prints, per kernel, an early (full) and a late (tail) dispatch."""
import csv, sys, collections
per = collections.OrderedDict()
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        key = (int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0][-40:])
        per.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
byk = collections.defaultdict(list)
for (did, name), c in per.items():
    byk[name].append((did, c))
for name, lst in byk.items():
    if not any(s in name for s in ("k_sweep", "k_riccati", "k_step")):
        continue
    lst.sort()
    for label, idx in (("early", min(4, len(lst) - 1)), ("late", len(lst) - 4 if len(lst) > 8 else len(lst) - 1)):
        did, c = lst[idx]
        print(name, label, "dispatch", did, {k: int(v) for k, v in sorted(c.items())})
