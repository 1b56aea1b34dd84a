"""Per-pass kernel durations of ONE solo batch from a rocprofv3 kernel trace CSV (development aid).
usage: pass_timeline.py <kernel_trace.csv>  -- prints pass, sweep/riccati/step durations (us)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [(r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, int(r["Start_Timestamp"])) for r in rows]
# last k_pack marks the start of the last batch
last = max(i for i, n in enumerate(names) if "k_pack" in n[0])
t0 = names[last][2]
p = 0
line = {}
for n, d, t in names[last:]:
    key = "sweep" if "k_sweep" in n else "ricc" if "k_riccati" in n else "step" if "k_step" in n else "comp" if "k_compact" in n else None
    if key is None:
        print(f"      {n[:40]:40s} {d:8.1f} us  t={(t - t0) / 1e3:9.1f}")
        continue
    if key == "sweep":
        if line:
            print(p, {k: round(v, 1) for k, v in line.items()})
        p += 1
        line = {"t": (t - t0) / 1e3}
    line[key] = line.get(key, 0.0) + d
print(p, {k: round(v, 1) for k, v in line.items()})
