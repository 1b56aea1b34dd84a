"""How many kernels of the solver run concurrently, and on which HSA queues (rocprofv3 kernel trace; development aid)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "rmpc::" in r["Kernel_Name"]]
print("columns:", list(rows[0].keys()))
q = collections.Counter((r.get("Queue_Id"), r.get("Stream_Id", "?"), r.get("Thread_Id", "?")) for r in rows)
for k, v in sorted(q.items()): print("queue/stream/thread", k, v)
ev = []
for r in rows:
    ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
t0 = ev[0][0]; t1 = ev[-1][0]
cur = 0; last = t0; hist = collections.Counter()
for t, d in ev:
    hist[cur] += t - last; last = t; cur += d
tot = sum(hist.values())
print("span ms", (t1 - t0) / 1e6, {k: round(v / tot, 3) for k, v in sorted(hist.items())})
