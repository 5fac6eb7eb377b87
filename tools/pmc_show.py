"""Per-kernel averages of the counters a tools/pmc_micro.sh run collected: python tools/pmc_show.py <dir> [kernel substring]"""
import csv, os, sys
from collections import defaultdict

root, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
acc = defaultdict(lambda: defaultdict(list))
for dp, _, fs in os.walk(root):
    for f in fs:
        if f.endswith("counter_collection.csv"):
            for r in csv.DictReader(open(os.path.join(dp, f))):
                if pat in r["Kernel_Name"]:
                    acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:28s} {sum(v) / len(v):16.0f}  (n={len(v)})")
