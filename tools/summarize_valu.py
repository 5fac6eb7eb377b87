#!/usr/bin/env python3
"""profiles/r02_valu_util.csv from the counter pass of tools/pmc_valu.sh: per kernel of the training step, the share of
the time in which the vector ALUs (and the LDS pipe) were issuing, by rocprofv3's own expressions evaluated per launch:
    VALUBusy = 100 * SQ_ACTIVE_INST_VALU * 4 / SIMD_NUM / GRBM_GUI_ACTIVE
    LDS issue = 100 * SQ_ACTIVE_INST_LDS * 4 / SIMD_NUM / GRBM_GUI_ACTIVE
(SQ counters are sums over the 8 XCDs, SIMD_NUM = 1024; GRBM_GUI_ACTIVE is summed over the XCDs too and divided by 8),
and the vector / LDS instructions per wave."""
import collections
import csv
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "pmc_valu", "train")


def short(name):
    m = re.match(r"(?:void )?\(anonymous namespace\)::(.+?)\(", name)
    return m.group(1) if m else name


def find(suffix):
    for dirpath, _, files in os.walk(SRC):
        for f in files:
            if f.endswith(suffix):
                return os.path.join(dirpath, f)
    raise SystemExit(f"no *{suffix} under {SRC}")


acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(find("counter_collection.csv"))):
    acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for k, c in acc.items():
    n = len(c["GRBM_GUI_ACTIVE"])
    avg = {name: sum(v) / len(v) for name, v in c.items()}
    gui = avg["GRBM_GUI_ACTIVE"] / 8.0
    if gui <= 0:
        continue
    waves = max(avg.get("SQ_WAVES", 0.0), 1.0)
    rows.append((k, n, round(gui), round(100 * avg["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / gui, 1),
                 round(100 * avg["SQ_ACTIVE_INST_LDS"] * 4 / 1024 / gui, 1), round(avg["SQ_INSTS_VALU"] / waves, 1),
                 round(avg["SQ_INSTS_LDS"] / waves, 1), int(waves)))
rows.sort(key=lambda r: -r[2] * r[1])
with open(os.path.join(ROOT, "profiles", "r02_valu_util.csv"), "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "launches", "gpu_cycles_per_launch", "valu_busy_pct", "lds_issue_pct", "valu_insts_per_wave",
                "lds_insts_per_wave", "waves"])
    w.writerows(rows)
for r in rows[:24]:
    print(r)
