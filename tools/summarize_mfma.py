#!/usr/bin/env python3
"""profiles/<tag>_mfma_util.csv (tag = argv[1], default r01) from the counter passes of tools/pmc_mfma.sh: per kernel, the MFMA pipe utilisation
(SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 1024 SIMDs), rocprofv3's MfmaUtil expression evaluated per launch; both
counters are sums over the 8 XCDs, and GRBM_GUI_ACTIVE is divided by 8 accordingly) and the MFMA flops per launch
(SQ_INSTS_VALU_MFMA_MOPS_* x 512)."""
import collections
import csv
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "pmc_mfma")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r01"


def short(name):
    m = re.match(r"(?:void )?\(anonymous namespace\)::(.+?)\(", name)
    return m.group(1) if m else name


rows = []
for wl in ("infer", "train"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(os.path.join(SRC, wl, f"{wl}_counter_collection.csv"))):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in acc.items():
        busy = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(c["SQ_VALU_MFMA_BUSY_CYCLES"])
        gui = sum(c["GRBM_GUI_ACTIVE"]) / len(c["GRBM_GUI_ACTIVE"])
        f32 = 512 * sum(c["SQ_INSTS_VALU_MFMA_MOPS_F32"]) / len(c["SQ_INSTS_VALU_MFMA_MOPS_F32"])
        b16 = 512 * sum(c["SQ_INSTS_VALU_MFMA_MOPS_BF16"]) / len(c["SQ_INSTS_VALU_MFMA_MOPS_BF16"])
        if busy == 0:
            continue
        rows.append((wl, k, len(c["GRBM_GUI_ACTIVE"]), round(busy), round(gui), round(100.0 * busy / (gui / 8.0 * 1024.0), 2),
                     int(f32), int(b16)))
rows.sort(key=lambda r: (r[0], -r[5]))
with open(os.path.join(ROOT, "profiles", f"{TAG}_mfma_util.csv"), "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["workload", "kernel", "launches", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "mfma_util_percent",
                "mfma_flops_f32_per_launch", "mfma_flops_bf16_per_launch"])
    w.writerows(rows)
for r in rows:
    print(r)
