#!/usr/bin/env python3
"""Build check: fails if a kernel whose waits for LDS-DMA transfers are COUNTED (s_waitcnt vmcnt(n), n = the number of
vector-memory instructions the wave has issued since the transfers it needs) uses scratch memory.  A spill adds
vector-memory instructions the count does not know about; the wait would then be too weak and a stage would be read
before its transfer has landed - silently wrong results, not a fault.
Usage: check_no_scratch.py <remarks file written by hipcc -Rpass-analysis=kernel-resource-usage> <regex over mangled kernel names>..."""
import re
import sys


def main():
    path, names = sys.argv[1], sys.argv[2:]
    text = open(path, errors="replace").read()
    blocks = re.split(r"remark: Function Name: ", text)[1:]
    seen, bad = set(), []
    for b in blocks:
        fn = b.split()[0]
        hit = [n for n in names if re.search(n, fn)]
        if not hit:
            continue
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b)
        if m is None:
            bad.append(f"{fn}: no ScratchSize remark")
            continue
        seen.update(hit)
        if int(m.group(1)) != 0:
            bad.append(f"{fn}: {m.group(1)} bytes of scratch per lane")
    bad += [f"{n}: no such kernel in {path}" for n in names if n not in seen]
    if bad:
        sys.exit("counted-wait kernels must not spill:\n  " + "\n  ".join(bad))


if __name__ == "__main__":
    main()
