#!/usr/bin/env python3
"""Turns the rocprofv3 output of tools/profile_r02.sh (gpurun_out/prof_r02/{train,infer}/) into the summaries kept under
profiles/: kernel stats, the bench line printed under the profiler, HBM bytes per launch per kernel (separate
FETCH_SIZE / WRITE_SIZE passes; FETCH_SIZE is in KB and counts half of the bytes of wide reads on gfx950, so
read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE in KB is exact).

Usage: python tools/summarize_prof.py [src_dir] [tag] [steps_in_pmc_pass]
       (defaults: gpurun_out/prof_r02/train, r02_train_b64_bf16s, 6; the last one only scales the printed per-step total)"""
import collections
import csv
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "prof_r02", "train")
TAG = sys.argv[2] if len(sys.argv) > 2 else "r02_train_b64_bf16s"
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 6
OUT = os.path.join(ROOT, "profiles")


def short(name: str) -> str:
    """'void (anonymous namespace)::k<1, 2>((anonymous namespace)::Args)' -> 'k<1, 2>'; mangled names stay."""
    m = re.match(r"(?:void )?\(anonymous namespace\)::(.+?)\(", name)
    return m.group(1) if m else name


def find(sub, suffix):
    for dirpath, _, files in os.walk(os.path.join(SRC, sub)):
        for f in files:
            if f.endswith(suffix):
                return os.path.join(dirpath, f)
    raise SystemExit(f"no *{suffix} under {SRC}/{sub}")


def counter_avg(sub, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(find(sub, "counter_collection.csv"))):
        if r["Counter_Name"] == counter:
            acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


shutil.copy(find("trace", "kernel_stats.csv"), os.path.join(OUT, f"{TAG}_kernel_stats.csv"))
shutil.copy(os.path.join(SRC, "bench_under_rocprof.json"), os.path.join(OUT, f"{TAG}_bench_under_rocprof.json"))
avg_ns = {short(r["Name"]): float(r["AverageNs"]) for r in csv.DictReader(open(find("trace", "kernel_stats.csv")))}
fetch, write = counter_avg("pmc_fetch", "FETCH_SIZE"), counter_avg("pmc_write", "WRITE_SIZE")
rows = []
for k, (f_kb, n) in fetch.items():
    w_kb = write.get(k, (0.0, 0))[0]
    rd, wr = 2.0 * f_kb * 1024.0, w_kb * 1024.0
    ns = avg_ns.get(k)
    rows.append((k, n, round(f_kb, 1), int(rd), round(w_kb, 1), int(wr), int(rd + wr), ns,
                 round((rd + wr) / ns, 1) if ns else None))
rows.sort(key=lambda r: -r[6])
with open(os.path.join(OUT, f"{TAG}_pmc_hbm.csv"), "w", newline="") as fh:
    wtr = csv.writer(fh)
    wtr.writerow(["kernel", "launches_in_pmc_pass", "FETCH_SIZE_KB_raw", "read_bytes_corrected", "WRITE_SIZE_KB", "write_bytes",
                  "hbm_bytes_per_launch", "avg_ns_kernel_trace", "hbm_GBps"])
    wtr.writerows(rows)
per_kernel = {r[0]: r[6] for r in rows}
# ksr_wgrad launches one of several wgrad_kernel instances followed by one reduce_kernel: launch-weighted average
wg = [r for r in rows if r[0].startswith(("wgrad_kernel<", "wgrad_fast_kernel<"))]
if wg:
    per_kernel["entry:ksr_wgrad"] = int(sum(r[6] * r[1] for r in wg) / sum(r[1] for r in wg) + next((v for k, v in per_kernel.items() if k.startswith("reduce_kernel")), 0))
# bench.py reads the default workload's file; other tags keep their own
with open(os.path.join(OUT, "pmc_hbm_bytes_per_launch.json" if TAG in ("r02_train_b64_bf16s", "r03_train_b64_bf16s") else f"{TAG}_pmc_hbm_bytes_per_launch.json"), "w") as fh:
    json.dump(per_kernel, fh, indent=1)
total = sum(r[6] * r[1] for r in rows)
print(f"wrote {len(rows)} kernels to profiles/{TAG}_*; HBM bytes over the PMC pass: {total / 1e9:.2f} GB "
      f"= {total / 1e9 / STEPS:.3f} GB per step over {STEPS} steps")
