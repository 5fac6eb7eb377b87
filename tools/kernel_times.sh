#!/bin/bash
# Per-kernel average durations of the headline training step (rocprofv3 kernel trace, one stream): tools/kernel_times.sh [grep pattern]
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
export KSR_SIDE_WGRAD=0
OUT=$R/gpurun_out/kt
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --headline-only ${KSR_BENCH_EXTRA} > $OUT/bench.json 2> $OUT/trace.log || exit 1
python3 - "$OUT" "$1" <<'PY'
import csv, os, re, sys
out, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
for dp, _, fs in os.walk(out):
    for f in fs:
        if f.endswith("kernel_stats.csv"):
            rows = list(csv.DictReader(open(os.path.join(dp, f))))
            steps = 12        # 5 timed + 2 warm-up + the 5 of the per-entry-point pass
            print(f"all kernels: {sum(float(r['TotalDurationNs']) for r in rows) / steps / 1e6:.3f} ms per step (if {steps} steps ran)")
            for r in rows:
                name = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
                name = re.sub(r"\(.*", "", name).replace("void ", "")
                if pat in name and float(r["Percentage"]) > 0.2:
                    print(f"{name[:70]:70s} calls {int(r['Calls']):4d} avg {float(r['AverageNs'])/1e3:8.1f} us  per step {float(r['TotalDurationNs'])/steps/1e3:7.1f} us  {float(r['Percentage']):5.2f} %")
PY
