#!/usr/bin/env python3
"""Summary of a tools/power_sample.sh / power_side.sh log: per label the samples taken while the kernel loop ran
(power above 1.5 x idle), median power and shader clock.  usage: python tools/power_summary.py log.txt"""
import re, sys, statistics as st
lab, rows = None, {}
for line in open(sys.argv[1]):
    m = re.match(r"== (\S+) ", line)
    if m:
        lab = m.group(1); rows.setdefault(lab, []).append([None, None]); continue
    if lab is None:
        continue
    m = re.search(r"Power \(W\): ([\d.]+)", line)
    if m and rows[lab][-1][0] is None:
        rows[lab][-1][0] = float(m.group(1))
    m = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", line)
    if m and rows[lab][-1][1] is None:
        rows[lab][-1][1] = int(m.group(1))
for lab, r in rows.items():
    r = [x for x in r if x[0] is not None and x[1] is not None]
    busy = [x for x in r if x[0] > 400.0]
    if not busy:
        print(f"{lab:28s} {len(r)} samples, none busy"); continue
    print(f"{lab:28s} busy samples {len(busy):3d}: power median {st.median(x[0] for x in busy):7.1f} W (max {max(x[0] for x in busy):.0f}), "
          f"sclk median {st.median(x[1] for x in busy)} MHz (min {min(x[1] for x in busy)}, max {max(x[1] for x in busy)})")
