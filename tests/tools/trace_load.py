"""Development aid: per-kernel durations and concurrency of a rocprofv3 kernel trace of several handles in flight.
usage: trace_load.py <kernel_trace.csv> [skip_ms]   (kernels starting before skip_ms after the first one are ignored)"""
import collections
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "rmpc::" in r["Kernel_Name"]]
t00 = min(int(r["Start_Timestamp"]) for r in rows)
skip = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 0.0
rows = [r for r in rows if int(r["Start_Timestamp"]) - t00 >= skip]
def short(n):
    n = n.split("rmpc::")[1]
    return n.split("<")[0].split("(")[0]
dur = collections.defaultdict(list)
ev = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    k = short(r["Kernel_Name"])
    dur[k].append((e - s) / 1e3)
    ev.append((s, 1, k)); ev.append((e, -1, k))
ev.sort()
span = (ev[-1][0] - ev[0][0]) / 1e6
print(f"span {span:.2f} ms, kernels {len(rows)}")
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    v = sorted(v)
    print(f"  {k:16s} n={len(v):5d} sum={sum(v) / 1e3:8.2f} ms  mean={sum(v) / len(v):7.1f} us  p10={v[len(v) // 10]:7.1f} p50={v[len(v) // 2]:7.1f} p90={v[9 * len(v) // 10]:7.1f}")
cur = collections.Counter(); last = ev[0][0]; hist = collections.Counter(); idle = 0
for t, d, k in ev:
    n = sum(cur.values())
    hist[n] += t - last
    last = t
    cur[k] += d
tot = sum(hist.values())
print("  concurrency (share of the span with n kernels running):", {k: round(v / tot, 3) for k, v in sorted(hist.items())})
