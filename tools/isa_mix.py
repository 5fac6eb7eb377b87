#!/usr/bin/env python3
"""Static instruction mix per kernel of one .hip file (gfx950 device assembly): python tools/isa_mix.py ksr_tx_bwd [filter]"""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "super-resolution_amd", "csrc", sys.argv[1] + ".hip")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = f"/tmp/{sys.argv[1]}.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=on", f"-I{ROOT}/include",
                       "-S", "--cuda-device-only", "-o", out, src], stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
cur, c = None, None
for l in lines:
    m = re.match(r"^(_Z\S+):\s", l)
    if m:
        cur, c = m.group(1), collections.Counter()
        continue
    if cur is None:
        continue
    t = l.strip()
    if t.startswith("s_endpgm"):
        if flt in cur:
            valu = sum(v for k, v in c.items() if k.startswith("v:"))
            top = sorted(((v, k) for k, v in c.items() if k.startswith("v:")), reverse=True)[:10]
            name = subprocess.run(["c++filt", cur], capture_output=True, text=True).stdout.strip()[:110]
            print(f"{name}\n   total {sum(c.values())} valu {valu} mfma {c['mfma']} lds {c['lds']} vmem {c['vmem']} salu {c['salu']} wait {c['wait']} | " +
                  " ".join(f"{k[2:]}={v}" for v, k in top))
        cur = None
        continue
    if not t or t.startswith((".", ";")) or t.endswith(":"):
        continue
    op = t.split()[0]
    if "mfma" in op: c["mfma"] += 1
    elif op.startswith("v_"): c["v:" + op.split("_")[1]] += 1
    elif op.startswith("ds_"): c["lds"] += 1
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): c["vmem"] += 1
    elif op.startswith("s_waitcnt"): c["wait"] += 1
    else: c["salu"] += 1
