#!/usr/bin/env python3
"""Prints the step time and the per-entry-point times of the last JSON line in a bench log (default gpurun_out/bench_quick.log)."""
import json
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/bench_quick.log"
lines = [x for x in open(path) if x.startswith("{")]
if not lines:
    sys.exit("no JSON line in " + path + ":\n" + "".join(open(path).readlines()[-15:]))
d = json.loads(lines[-1])
print("ms_per_step", d["ms_per_step"], "value", d["value"], "| dominant", d["roofline"]["kernel"], d["roofline"]["frac"])
ep = d["roofline"]["ms_per_step_by_entry_point"]
print(" ".join(f"{k[4:]}={v:.3f}" for k, v in sorted(ep.items(), key=lambda kv: -kv[1])))
print("sum of entry points %.3f ms" % sum(ep.values()))
if d.get("inference"):
    print("inference", d["inference"]["ms_per_frame"], "ms/frame")
