#!/usr/bin/env python3
"""Reads a rocprofv3 kernel trace of a training workload (tools/b8_trace.py, bench.py --workload ...) and prints, for the
chosen step (adam_kernel to adam_kernel; default the last one, or the k-th from the end), every launch with its queue, its
duration and the time the GPU sat idle before it, then the totals and the time per kernel name; the spans of all steps come
first: python tools/step_timeline.py <dir with *kernel_trace.csv> [k]"""
import csv
import os
import sys

rows = []
for dp, _, fs in os.walk(sys.argv[1]):
    for f in fs:
        if f.endswith("kernel_trace.csv"):
            rows += list(csv.DictReader(open(os.path.join(dp, f))))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
print("step spans (us):", " ".join(f"{(int(rows[adam[i + 1]]['End_Timestamp']) - int(rows[adam[i]]['End_Timestamp'])) / 1e3:.0f}"
                                  for i in range(len(adam) - 1)))
kth = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lo, hi = adam[-kth - 1] + 1, adam[-kth] + 1
step = rows[lo:hi]
t0 = int(step[0]["Start_Timestamp"])
span = int(step[-1]["End_Timestamp"]) - int(rows[lo - 1]["End_Timestamp"])
busy, prev_end = 0, int(rows[lo - 1]["End_Timestamp"])
cover_end = prev_end
print(f"{len(step)} launches, step span {span / 1e3:.1f} us")
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:58]
    gap = s - cover_end
    print(f"{(s - t0) / 1e3:9.1f} us  q{r.get('Queue_Id', '?'):>2} dur {(e - s) / 1e3:7.1f}  idle before {max(gap, 0) / 1e3:6.1f}  {name}")
    busy += max(0, e - max(s, cover_end))
    cover_end = max(cover_end, e)
import collections
per = collections.defaultdict(lambda: [0, 0])
for r in step:
    n = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:70]
    per[n][0] += 1
    per[n][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for n, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print(f"   {t / 1e3:8.1f} us  x{c:3d}  avg {t / c / 1e3:6.1f}  {n}")
print(f"GPU busy (union of kernels) {busy / 1e3:.1f} us of {span / 1e3:.1f} us; idle {(span - busy) / 1e3:.1f} us")
